#!/bin/bash
# One bench.py line per workload (the judged JSON of each), run on the GPU box:
#   gpurun -- 'bash tools/bench_all.sh r02_p'   -> gpurun_out/<tag>_bench_*.json   (copy into profiles/)
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export GPU_MAX_HW_QUEUES=8
cd "$ROOT"
python3 bench.py > "$OUT/${TAG}_bench_c2_f32.json" 2>/dev/null
python3 bench.py --s-val 220 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_s220.json" 2>/dev/null
python3 bench.py --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_bf16.json" 2>/dev/null
python3 bench.py --config C3 --dtype f32 --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c3_f32.json" 2>/dev/null
python3 bench.py --config C3 --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c3_bf16.json" 2>/dev/null
python3 bench.py --config C4 --steps 30 --warmup 10 --cpu-rays 256 --cpu-iters 1 > "$OUT/${TAG}_bench_c4_lts_f32.json" 2>/dev/null
python3 bench.py --config C4 --stage pdra --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c4_pdra_f32.json" 2>/dev/null
python3 bench.py --config C5 --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c5_pdra_bf16.json" 2>/dev/null
python3 bench.py --config C5 --stage finetune --steps 30 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c5_finetune_bf16.json" 2>/dev/null
python3 bench.py --grid 256 --steps 30 --warmup 8 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_g256_f32.json" 2>/dev/null
python3 bench.py --oblique --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_oblique_f32.json" 2>/dev/null
python3 bench.py --grid 256 --oblique --steps 30 --warmup 8 --no-cpu-baseline > "$OUT/${TAG}_bench_c2_g256_oblique_f32.json" 2>/dev/null
for f in "$OUT/${TAG}"_bench_*.json; do echo "$(basename $f): $(tail -1 $f | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['dtype'], d['roofline'].get('frac'), (d['roofline'].get('whole_step') or {}).get('frac'))")"; done
