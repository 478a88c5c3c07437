#!/bin/bash
# per-kernel time of the default bench workload, kernels serialised (bench.py --serial):
#   gpurun -- 'bash tools/kstats.sh <tag> [bench args]'   ->  gpurun_out/<tag>_stats/run_kernel_stats.csv + a table on stdout
TAG=${1:-k}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GPU_MAX_HW_QUEUES=8       # as bench.py sets it; under rocprofv3 HIP is initialised before python runs, so it must come from the shell
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_stats" -o run -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-optimizer --no-kernel-timing --serial "$@" > "$ROOT/gpurun_out/${TAG}_stats.log" 2>&1
cd "$ROOT"
find "gpurun_out/${TAG}_stats" -type f ! -name "*kernel_stats.csv" -delete
python3 tools/kstats_table.py "gpurun_out/${TAG}_stats/run_kernel_stats.csv" 13
