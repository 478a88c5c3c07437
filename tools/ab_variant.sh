#!/bin/bash
# A/B of the wave-pair kernels inside the real step, alternating on ONE box:  gpurun -- 'bash tools/ab_variant.sh [rounds] [bench args]'
# prints ms/step of bench.py --split-variant 0 / 2 / 1 per round (C2 f32 unless bench args say otherwise)
ROUNDS=${1:-3}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for r in $(seq 1 $ROUNDS); do
    for v in 0 2 1; do
        ms=$(python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other --no-kernel-timing --split-variant $v "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
        echo "round $r variant $v: $ms ms/step"
    done
done
