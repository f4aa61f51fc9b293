#!/usr/bin/env python
"""HBM traffic of the dominant kernel (the bulk trailing update, rank 512 at K = 1000) from
two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (section HBM): both
counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of
wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact.  (The
C-tile reads of this kernel are 8 B per lane, a width the guide lists as
uncalibrated; the doubled figure is therefore an upper estimate of the read
side.)  Bulk launches = the gemm_nt_kernel dispatches with the lower-triangle
grid of the trailing update: at K = 1000 with super-panels of 8 there is one per
step, n = 512 (36 lower tiles x 64 stars), a grid no block-column update has.
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    import os
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'gemm_nt_kernel' not in r['Kernel_Name'] or r['Counter_Name'] != counter:
            continue
        out[r['Dispatch_Id']] = (int(r['Grid_Size']), float(r['Counter_Value']))
    return out


fetch = load(sys.argv[1], 'FETCH_SIZE')
write = load(sys.argv[2], 'WRITE_SIZE')
# bulk grid for S = 64 stars, K = 1000 (Kp = 1024), super-panels of 8: trailing size
# n = 512 -> 36 lower tiles x 64 stars x 256 threads (block-column updates have <= 15 x 64)
grids = [36 * 64 * 256]


def bulk_only(d):
    return [v for g, v in d.values() if g in grids]


sel_f = bulk_only(fetch)
sel_w = bulk_only(write)
n = min(len(sel_f), len(sel_w))
fetch_b = 2.0 * 1024.0 * sum(sel_f[:n]) / n
write_b = 1024.0 * sum(sel_w[:n]) / n
res = {
    "kernel": "gemm_nt_kernel (bulk rank-512 trailing update of n = 512, lower tiles)",
    "launches_averaged": n,
    "grid_sizes": grids,
    "FETCH_SIZE_KiB_per_launch_raw": sum(sel_f[:n]) / n,
    "WRITE_SIZE_KiB_per_launch": sum(sel_w[:n]) / n,
    "fetch_bytes_per_launch_corrected_x2": fetch_b,
    "write_bytes_per_launch": write_b,
    "gemm_nt_bytes_per_launch": fetch_b + write_b,
}
json.dump(res, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(res, indent=1))
