#!/usr/bin/env python
"""One K = 1000, 64-star step captured in a HIP graph (torch.cuda.CUDAGraph) against the same step
enqueued launch by launch: does a graph shorten the gaps between the ~35 dependent launches?"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from chain_check import engine, setup

for panel in (0, 1):
    e = engine(0, panel)
    a = setup(e, 64, 1000)
    ws = e.workspace(64, 1000, 1)
    out = e.empty(64)
    def step():
        e.lnlike_ensemble(a[0], a[1], a[2], tab=a[3], meanvar=a[4], out=out, workspace=ws)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ref = out.clone()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200 * 1e3
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.synchronize()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 200 * 1e3
    print("panel mode %d: launch by launch %.4f ms, graph replay %.4f ms, same values %s" % (
        panel, eager, graph, bool(torch.equal(ref, out))), flush=True)
