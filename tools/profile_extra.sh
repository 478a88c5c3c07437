#!/bin/bash
# rocprofv3 kernel-trace summaries of the non-headline workloads (run on the GPU box):
#   gpurun -- 'bash tools/profile_extra.sh r01_f'
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-optimizer --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_c4lts" -o run -- python3 "$ROOT/bench.py" --config C4 --steps 8 --warmup 3 $ARGS > "$OUT/${TAG}_c4lts.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_c2bf16" -o run -- python3 "$ROOT/bench.py" --dtype bf16 --steps 10 --warmup 3 $ARGS > "$OUT/${TAG}_c2bf16.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_c2s220" -o run -- python3 "$ROOT/bench.py" --s-val 220 --steps 10 --warmup 3 $ARGS > "$OUT/${TAG}_c2s220.log" 2>&1
for w in c4lts c2bf16 c2s220; do find "$OUT/${TAG}_$w" -type f ! -name "*kernel_stats.csv" -delete; done
for w in c4lts c2bf16 c2s220; do tail -n 1 "$OUT/${TAG}_$w.log" | cut -c1-200; done
cd "$ROOT"
python3 bench.py --config C4 --steps 20 --warmup 5 --cpu-rays 256 --cpu-iters 1 > "$OUT/${TAG}_bench_c4_lts.json" 2>/dev/null
python3 bench.py --config C4 --stage pdra --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_bench_c4_pdra.json" 2>/dev/null
python3 bench.py --s-val 220 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_s220.json" 2>/dev/null
python3 bench.py --config C3 --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c3.json" 2>/dev/null
python3 bench.py --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_bf16.json" 2>/dev/null
python3 bench.py --config C3 --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c3_bf16.json" 2>/dev/null
python3 bench.py --config C4 --dtype bf16 --stage pdra --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_bench_c5_pdra_bf16.json" 2>/dev/null
ls "$OUT" | head -40
