#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel name.
python tools/pmc.py <dir> [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
import os
f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if sub and sub not in k:
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r['Dispatch_Id'])
for k in agg:
    n = len(disp[k])
    print(k, "dispatches", n)
    for c, v in sorted(agg[k].items()):
        print("   %-32s total %.6g   per-dispatch %.6g" % (c, v, v / n))
