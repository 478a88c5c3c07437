#!/bin/bash
# rocprofv3 evidence for ONE bench.py workload, run on the GPU box:
#   gpurun -- 'bash tools/profile_workload.sh <tag> [bench.py args, e.g. --config C3 --dtype bf16]'
# 1. --kernel-trace --stats  -> gpurun_out/<tag>_stats/   (per-kernel time, kernels serialised: bench.py --serial)
# 2. --pmc FETCH_SIZE        -> gpurun_out/<tag>_fetch/   (separate pass per counter, as the guide prescribes)
# 3. --pmc WRITE_SIZE        -> gpurun_out/<tag>_write/
# The program goes directly after `--` (no env/bash wrappers under the profiler).  tools/pmc_summary.py <tag> [same bench
# args] then writes profiles/<tag>_* and merges the workload's HBM bytes into profiles/pmc_traffic.json.
TAG=${1:-prof}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export GPU_MAX_HW_QUEUES=8       # as bench.py sets it; under rocprofv3 HIP is initialised before python runs, so it must come from the shell
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-optimizer --no-kernel-timing --serial $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o run -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 $ARGS > "$OUT/${TAG}_stats.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 $ARGS > "$OUT/${TAG}_fetch.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 $ARGS > "$OUT/${TAG}_write.log" 2>&1
for w in stats fetch write; do echo "== $w"; tail -n 1 "$OUT/${TAG}_$w.log" | cut -c1-200; done
# per-(kernel, grid size) table from the per-dispatch trace (before the trace is dropped): what bench.py's single-launch
# roofline figure can be checked against
python3 "$ROOT/tools/kstats_by_grid.py" "$(find "$OUT/${TAG}_stats" -name "*kernel_trace.csv" | head -1)" "$OUT/${TAG}_stats/kernel_by_grid.csv" 13
# keep the csv summaries only (gpurun copies back at most 64 MiB)
find "$OUT" -path "*${TAG}_*" -type f ! -name "*.csv" ! -name "*.log" -delete
find "$OUT" -path "*${TAG}_*" -name "*kernel_trace.csv" -delete
find "$OUT" -path "*${TAG}_*" -name "*.csv" -size +20M -delete
cd "$ROOT" && python3 tools/kstats_table.py "$OUT/${TAG}_stats/run_kernel_stats.csv" 13 14
