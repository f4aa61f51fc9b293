"""Per-launch durations of one likelihood step from a rocprofv3 --kernel-trace CSV: the launches
of the last step in stream order, with their grid sizes (which panel / super-panel they are).
usage: python tools/trace_step.py <kernel_trace.csv> [launches_per_step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from a polar_moments launch to the next
idx = [i for i, r in enumerate(rows) if "polar_moments" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:28]
    print("%-28s grid %8s  start %8.1f  dur %6.1f  gap %5.1f" % (
        name, int(r["Grid_Size_X"]) // 256, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
    tot += (e - s) / 1e3
print("step span %.1f us, kernel time %.1f us" % ((prev_end - t0) / 1e3, tot))
