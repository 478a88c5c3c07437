#!/usr/bin/env python3
"""Per-kernel counter sums of one tools/pmc_pass.sh run (dispatches of the LAST profiled step):
    python tools/pmc_kernel.py <tag> [kernel-name substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
G = os.path.join(ROOT, "gpurun_out", tag)
cc = list(csv.DictReader(open(glob.glob(os.path.join(G, "**/*counter_collection.csv"), recursive=True)[0])))
kt = list(csv.DictReader(open(glob.glob(os.path.join(G, "**/*kernel_trace.csv"), recursive=True)[0])))
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt}
disp, order = {}, []
for r in cc:
    d = r["Dispatch_Id"]
    if d not in disp:
        disp[d] = dict(name=r["Kernel_Name"], ns=dur.get(d, 0))
        order.append(d)
    disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [disp[d] for d in order if "anonymous namespace" in disp[d]["name"]]
starts = [i for i, r in enumerate(rows) if "march_kernel<0" in r["name"]]
rows = rows[starts[-1]:] if starts else rows
agg = defaultdict(lambda: defaultdict(float))
for r in rows:
    n = r["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if sub in n:
        agg[n]["launches"] += 1
        for k, v in r.items():
            if k != "name":
                agg[n][k] += v
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    print(n, {k: (int(v) if k != "ns" else round(v / 1e6, 4)) for k, v in a.items()})
