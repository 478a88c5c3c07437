#!/usr/bin/env python3
"""Per-(kernel, grid size) statistics of a rocprofv3 --kernel-trace run: the --stats table groups every dispatch of a
kernel SYMBOL, which hides launches of the same symbol with different work (the radiance net's and the tone mapper's
share one template family; round 2's two forward launches shared one instantiation).  bench.py's roofline.frac is the
figure of ONE named launch: this table lets it be re-derived from the committed profile.
    python tools/kstats_by_grid.py <run_kernel_trace.csv> <out.csv> [steps]"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
agg = defaultdict(list)
for r in csv.DictReader(open(src)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
    agg[(name, grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_size", "workgroup_size", "calls", "calls_per_step", "avg_us", "min_us", "max_us", "total_ms"])
    for (name, grid, wg), d in rows:
        if sum(d) < 1e-4 * sum(sum(v) for v in agg.values()):
            continue
        w.writerow([name, grid, wg, len(d), f"{len(d) / steps:.2f}" if steps else "", f"{sum(d) / len(d) / 1e3:.2f}",
                    f"{min(d) / 1e3:.2f}", f"{max(d) / 1e3:.2f}", f"{sum(d) / 1e6:.3f}"])
print("wrote", dst, len(rows), "rows")
