#!/usr/bin/env python3
"""GPU idle gaps of the last profiled step in a rocprofv3 kernel trace:  python tools/trace_gaps.py <dir-with-kernel_trace.csv> [min_gap_us]"""
import csv, glob, os, sys
d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
f = glob.glob(os.path.join(d, "**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "march_kernel<0" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
busy_end = t0
print(f"step wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {len(step)} dispatches")
for r in step + [rows[b]]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - busy_end > min_gap * 1e3:
        print(f"  gap {(s - busy_end) / 1e3:7.1f} us at +{(busy_end - t0) / 1e3:8.1f} us before {r['Kernel_Name'][:70]}")
    busy_end = max(busy_end, e)
