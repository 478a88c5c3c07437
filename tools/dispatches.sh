#!/bin/bash
# Dispatches per step of one bench.py workload (every kernel / copy / fill the device executes, from a rocprofv3 kernel
# trace with the kernels serialised):   gpurun -- 'bash tools/dispatches.sh <tag> [bench args, e.g. --config C5]'
#   -> gpurun_out/<tag>_dispatches.txt (per kernel: launches per step, microseconds per step) + the per-(kernel, grid) csv
TAG=${1:-d}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export GPU_MAX_HW_QUEUES=8
STEPS=10; WARM=3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/${TAG}_trace" -o run -- python3 "$ROOT/bench.py" --steps $STEPS --warmup $WARM --no-cpu-baseline --no-optimizer --no-kernel-timing --serial --no-other "$@" > "$OUT/${TAG}_trace.log" 2>&1
cd "$ROOT"
TR=$(find "$OUT/${TAG}_trace" -name "*kernel_trace.csv" | head -1)
python3 tools/kstats_by_grid.py "$TR" "$OUT/${TAG}_kernel_by_grid.csv" $((STEPS + WARM))
python3 - "$TR" $((STEPS + WARM)) > "$OUT/${TAG}_dispatches.txt" <<'PY'
import collections, csv, sys
steps = int(sys.argv[2])
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[0] for a in agg.values()) / steps
print(f"dispatches per step (all kernels, copies and fills; {steps} steps incl. warm-up): {tot:.1f}   "
      f"kernel time per step: {sum(a[1] for a in agg.values()) / steps / 1e3:.3f} ms")
print(f"of them shorter than 10 us on average: {sum(a[0] for a in agg.values() if a[1] / a[0] < 10.0) / steps:.1f}")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{c / steps:7.2f} per step {us / steps:9.1f} us per step  {k}")
PY
head -40 "$OUT/${TAG}_dispatches.txt"
rm -rf "$OUT/${TAG}_trace"
