#!/usr/bin/env python
"""HBM traffic of a whole likelihood step, per kernel, from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    python tools/step_traffic.py <fetch_dir> <write_dir> <steps> [out.json]

Both counters are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: it reports half the
bytes of wide coalesced reads; an upper estimate for narrower accesses), WRITE_SIZE is exact.
Infinity-Cache hits are included in both (the counters sit on the L2's memory side)."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    import os
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
        out[k][0] += 1
        out[k][1] += float(r['Counter_Value'])
    return out


steps = float(sys.argv[3])
fetch = load(sys.argv[1], 'FETCH_SIZE')
write = load(sys.argv[2], 'WRITE_SIZE')
rows = []
for k in sorted(set(fetch) | set(write)):
    fb = 2.0 * 1024.0 * fetch[k][1] / steps
    wb = 1024.0 * write[k][1] / steps
    rows.append((fb + wb, k, fetch[k][0] / steps, fb, wb))
rows.sort(reverse=True)
tot_f = sum(r[3] for r in rows)
tot_w = sum(r[4] for r in rows)
for t, k, n, fb, wb in rows[:14]:
    print("%-60s launches/step %5.1f  read %8.1f MB  written %8.1f MB" % (k, n, fb / 1e6, wb / 1e6))
print("whole step: read %.1f MB + written %.1f MB = %.2f GB  (algorithmic 64 x 24 K^2 = 1.54 GB at K = 1000)" % (
    tot_f / 1e6, tot_w / 1e6, (tot_f + tot_w) / 1e9))
if len(sys.argv) > 4:
    json.dump({"read_bytes_per_step": tot_f, "written_bytes_per_step": tot_w,
               "per_kernel": [{"kernel": k, "launches_per_step": n, "read_bytes": fb, "written_bytes": wb}
                              for t, k, n, fb, wb in rows]}, open(sys.argv[4], "w"), indent=1)
