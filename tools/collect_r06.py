#!/usr/bin/env python
"""Copy the judged summaries of gpurun_out/r06 (tools/profile_r06.sh) into profiles/ and derive the
per-step HBM traffic file bench.py quotes (`roofline.traffic`).
python tools/collect_r06.py gpurun_out/r06"""
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


fn = os.path.join(O, "driver_cmd_bench.json")
if os.path.exists(fn):
    for l in open(fn):
        if l.startswith("{"):
            json.dump(json.loads(l), open(os.path.join(P, "r06_driver_cmd_bench.json"), "w"), indent=1)
copy(last("stats_inflight/**/*kernel_stats.csv"), "r06_inflight_kernel_stats.csv")
copy(last("stats_one/**/*kernel_stats.csv"), "r06_one_kernel_stats.csv")
copy(last("stats_cfg5/**/*kernel_stats.csv"), "r06_cfg5_kernel_stats.csv")
copy(last("stats_samples/**/*kernel_stats.csv"), "r06_samples_kernel_stats.csv")
for name in ("one_step_trace.txt", "prof_elp.txt", "inflight_probe.txt", "grad_timing.txt", "cfg5_shape.txt",
             "pmc_lds.txt", "stats_unplanned_inflight.txt", "mfma_budget.txt", "samples_bench.txt", "k_sweep.txt", "small_k_probe.txt", "k_sweep_blocked_small.txt",
             "cfg5_one_step_trace.txt"):
    copy(os.path.join(O, name), "r06_" + name)


def steps_of(d):   # steps in the PMC runs: the planned assembly's dispatches (one per step)
    f = last(d + "/**/*kernel_trace.csv")
    return sum(1 for r in csv.DictReader(open(f)) if "assemble_planned_kernel" in r["Kernel_Name"]) if f else 0


fd, wd = os.path.join(O, "pmc_FETCH_SIZE"), os.path.join(O, "pmc_WRITE_SIZE")
if os.path.isdir(fd) and os.path.isdir(wd):
    n = steps_of("pmc_FETCH_SIZE")
    out = os.path.join(P, "r06_step_traffic.json")
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_traffic.py"), fd, wd, str(n), out],
                         capture_output=True, text=True).stdout
    open(os.path.join(P, "r06_step_traffic.txt"), "w").write(txt)
    tr = json.load(open(out))
    # (the panel kernel is one instantiation per kind of launch since round 4: all of them together)
    dom = [k for k in tr["per_kernel"] if k["kernel"].startswith("panel_kernel")]
    if dom:
        tr["dominant_kernel"] = "panel_kernel<...> (%d instantiations)" % len(dom)
        tr["dominant_kernel_bytes_per_launch"] = (sum(k["read_bytes"] + k["written_bytes"] for k in dom) /
                                                  sum(k["launches_per_step"] for k in dom))
    tr["steps_profiled"] = n
    tr["note"] = ("FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM: an upper estimate for accesses narrower than 16 B per "
                  "lane) + WRITE_SIZE, separate --pmc passes, one step at a time")
    json.dump(tr, open(out, "w"), indent=1)
txt = ""
for d in ("pmc_sq", "pmc_grbm"):
    if last(d + "/**/*counter_collection.csv"):
        txt += subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc.py"), os.path.join(O, d)],
                              capture_output=True, text=True).stdout
if txt:
    open(os.path.join(P, "r06_pmc_sq.txt"), "w").write(txt)
print("profiles/:", sorted(f for f in os.listdir(P) if f.startswith("r06_")))
