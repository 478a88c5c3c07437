#!/bin/bash
# One rocprofv3 counter pass over bench.py (kernel trace + the given counters), raw csv kept under gpurun_out/<tag>/:
#   gpurun -- 'bash tools/pmc_pass.sh tagname "SQ_WAVE_CYCLES SQ_INSTS_LDS ..." [bench args]'
# Read with: python tools/pmc_kernel.py tagname [kernel-substring]
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8       # as bench.py sets it; under rocprofv3 HIP is initialised before python runs, so it must come from the shell
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/$TAG" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-optimizer --no-kernel-timing --serial "$@" > "$OUT/$TAG.log" 2>&1
tail -n 1 "$OUT/$TAG.log" | cut -c1-200
find "$OUT/$TAG" -type f ! -name "*.csv" -delete
find "$OUT/$TAG" -name "*.csv" -size +20M -delete
