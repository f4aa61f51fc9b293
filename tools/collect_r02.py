#!/usr/bin/env python
"""Copy the judged summaries of gpurun_out/r02 (tools/profile_r02.sh) into profiles/ and derive the
per-step HBM traffic file bench.py quotes (`roofline.traffic`).
python tools/collect_r02.py gpurun_out/r02"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
O = sys.argv[1]
P = os.path.join(ROOT, "profiles")


def last(pattern):
    f = sorted(glob.glob(os.path.join(O, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def copy(src, dst):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))


# bench line of the driver's command
line = None
fn = os.path.join(O, "driver_cmd_bench.json")
if os.path.exists(fn):
    for l in open(fn):
        if l.startswith("{"):
            line = json.loads(l)
    if line:
        json.dump(line, open(os.path.join(P, "r02_driver_cmd_bench.json"), "w"), indent=1)
for name in ("stats_inflight", "stats_one", "stats_one_onelaunch"):
    copy(last(name + "/**/*kernel_stats.csv"), "r02_%s_kernel_stats.csv" % name.replace("stats_", ""))
for name in ("mm_bench.txt", "strip_bench.txt", "check_modes.txt", "chain_trace.txt"):
    copy(os.path.join(O, name), "r02_" + name)
for src, dst in (("recursive_bench.json", "r02_recursive_driver_bench.json"),
                 ("dataflow_bench.json", "r02_dataflow_driver_bench.json")):
    fn = os.path.join(O, src)
    if os.path.exists(fn):
        for l in open(fn):
            if l.startswith("{"):
                json.dump(json.loads(l), open(os.path.join(P, dst), "w"), indent=1)

# launches per step in the PMC runs: count the lnlike_reduce dispatches (one per step)
def steps_of(d):
    f = last(d + "/**/*kernel_trace.csv")
    return sum(1 for r in csv.DictReader(open(f)) if "lnlike_reduce_kernel" in r["Kernel_Name"]) if f else 0


fd, wd = os.path.join(O, "pmc_FETCH_SIZE"), os.path.join(O, "pmc_WRITE_SIZE")
if os.path.isdir(fd) and os.path.isdir(wd):
    n = steps_of("pmc_FETCH_SIZE")
    out = os.path.join(P, "r02_step_traffic.json")
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_traffic.py"), fd, wd, str(n), out],
                         capture_output=True, text=True).stdout
    open(os.path.join(P, "r02_step_traffic.txt"), "w").write(txt)
    tr = json.load(open(out))
    dom = [k for k in tr["per_kernel"] if k["kernel"].startswith("gemm_nt_kernel<32, false, 2, 0, true>")]
    if dom:
        tr["dominant_kernel"] = dom[0]["kernel"]
        tr["dominant_kernel_bytes_per_launch"] = (dom[0]["read_bytes"] + dom[0]["written_bytes"]) / dom[0]["launches_per_step"]
    tr["steps_profiled"] = n
    tr["note"] = ("FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM: an upper estimate for accesses narrower than 16 B per "
                  "lane) + WRITE_SIZE, separate --pmc passes, one step at a time, one-launch-per-panel mode")
    json.dump(tr, open(out, "w"), indent=1)
sq = last("pmc_sq/**/*counter_collection.csv")
if sq:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc.py"), os.path.join(O, "pmc_sq")],
                         capture_output=True, text=True).stdout
    open(os.path.join(P, "r02_pmc_sq.txt"), "w").write(txt)
print("profiles/:", sorted(f for f in os.listdir(P) if f.startswith("r02_")))
