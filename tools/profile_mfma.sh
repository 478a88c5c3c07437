#!/bin/bash
# Matrix-core utilisation and effective clock of every kernel of bench.py's default workload (C2):
#   gpurun -- 'bash tools/profile_mfma.sh r01_g'
# Two counter passes (rocprofv3 --pmc, with --kernel-trace only), summarised by tools/mfma_summary.py:
#   pass A: GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
#   pass B: SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32
# The program goes directly after `--` (no env/bash wrappers under the profiler).
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8       # as bench.py sets it; under rocprofv3 HIP is initialised before python runs, so it must come from the shell
ARGS="--no-cpu-baseline --no-optimizer --no-kernel-timing --serial ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/${TAG}_mfmaA" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 $ARGS > "$OUT/${TAG}_mfmaA.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d "$OUT/${TAG}_mfmaB" -o run -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 $ARGS > "$OUT/${TAG}_mfmaB.log" 2>&1
for w in mfmaA mfmaB; do echo "== $w"; tail -n 2 "$OUT/${TAG}_$w.log" | cut -c1-300; done
find "$OUT" -path "*${TAG}_mfma*" -type f ! -name "*.csv" ! -name "*.log" -delete
find "$OUT" -path "*${TAG}_mfma*" -name "*.csv" -size +20M -delete
find "$OUT" -path "*${TAG}_mfma*" -name "*.csv" | head -20
