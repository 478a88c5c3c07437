#!/usr/bin/env python3
"""Every dispatch of the last profiled step of a rocprofv3 kernel trace, in start order, with the idle time in front of it:
   python tools/trace_step.py <dir-with-kernel_trace.csv> [marches-per-step, default 1; 2 for the lts / pdra stages]"""
import csv, glob, os, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "march_kernel<0" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = starts[-1 - k], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
busy_end = t0
idle = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - busy_end) / 1e3
    idle += gap
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"+{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  idle-before {gap:6.1f}  q{r.get('Queue_Id', '?'):>3}  {name[:80]}")
    busy_end = max(busy_end, e)
print(f"step wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, idle {idle:.1f} us")
