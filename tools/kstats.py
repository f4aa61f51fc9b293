#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats output directory:
python tools/kstats.py <dir> [steps]"""
import csv, glob, sys
import os
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    name = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:34]
    print("%-34s calls %5s total %8.3f ms  avg %8.1f us  per-step %7.1f us  %5.1f%%" % (
        name, r['Calls'], int(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
        int(r['TotalDurationNs']) / 1e3 / steps, float(r['Percentage'])))
print("total kernel ms %.3f ; per step %.3f ms" % (tot / 1e6, tot / 1e6 / steps))
