#!/usr/bin/env python3
"""Summarise tools/profile_mfma.sh:  python tools/mfma_summary.py <tag>   -> profiles/<tag>_mfma_util.csv

Per kernel symbol (our kernels, dispatches of the LAST profiled step averaged per symbol):
  ms            dispatch duration from the kernel trace of the same pass
  clock_ghz     GRBM_GUI_ACTIVE / 8 XCDs / duration   (MI355X_MICROARCH.md, DVFS give-back; reads high below ~0.3 ms)
  mfma_busy     SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)   -- the guide's MfmaUtil
  valu_active   SQ_ACTIVE_INST_VALU * 4 / SQ_WAVE_CYCLES / 4 ... reported as quad-cycle ratio of wave time
  wait_any      SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of wave time spent waiting to issue)
  mfma_gflop    SQ_INSTS_VALU_MFMA_MOPS_F32 * 512 / 1e9 (issued, padding included)
"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
N_SIMD, N_XCD = 1024, 8


def one(pattern):
    hits = glob.glob(os.path.join(G, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"no file matches {pattern}")
    return hits[0]


def load(which):
    cc = list(csv.DictReader(open(one(f"{tag}_{which}/**/*counter_collection.csv"))))
    kt = list(csv.DictReader(open(one(f"{tag}_{which}/**/*kernel_trace.csv"))))
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in kt}
    disp = {}
    order = []
    for r in cc:
        d = r["Dispatch_Id"]
        if d not in disp:
            disp[d] = dict(name=r["Kernel_Name"], ns=dur.get(d, 0))
            order.append(d)
        disp[d][r["Counter_Name"]] = disp[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [disp[d] for d in order if "anonymous namespace" in disp[d]["name"]]
    starts = [i for i, r in enumerate(rows) if "march_kernel<0" in r["name"]]
    return rows[starts[-1]:] if starts else rows


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


agg = defaultdict(lambda: defaultdict(float))
for which in ("mfmaA", "mfmaB"):
    for r in load(which):
        a = agg[short(r["name"])]
        a[f"n_{which}"] += 1
        a[f"ns_{which}"] += r["ns"]
        for k, v in r.items():
            if k not in ("name", "ns"):
                a[k] += v

os.makedirs(P, exist_ok=True)
out = os.path.join(P, f"{tag}_mfma_util.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches_per_step", "ms_per_launch", "clock_ghz", "mfma_busy_frac", "cu_busy_frac",
                "wait_any_frac_of_wave_cycles", "valu_active_frac_of_wave_cycles", "mfma_issued_gflop_per_launch"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns_mfmaA"]):
        n = max(a["n_mfmaA"], 1)
        ns = a["ns_mfmaA"]
        cyc = a["GRBM_GUI_ACTIVE"] / N_XCD
        clock = cyc / ns if ns else 0
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_SIMD) if cyc else 0
        cu = a["SQ_BUSY_CU_CYCLES"] / (cyc * 256) if cyc else 0
        wc = a["SQ_WAVE_CYCLES"]
        w.writerow([k, int(n), f"{ns / n / 1e6:.4f}", f"{clock:.3f}", f"{busy:.3f}", f"{cu:.3f}",
                    f"{a['SQ_WAIT_INST_ANY'] / wc:.3f}" if wc else "", f"{a['SQ_ACTIVE_INST_VALU'] / wc:.3f}" if wc else "",
                    f"{a['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / max(a['n_mfmaB'], 1) / 1e9:.3f}"])
print(open(out).read())
