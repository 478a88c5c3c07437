#!/bin/bash
# A/B of two builds of the library, per kernel (bench.py's instrumented pass), alternating on ONE box:
#   gpurun -- 'bash tools/ab_kernels.sh tools/_variants/X.so [rounds] [bench args]'
VAR=$1; ROUNDS=${2:-2}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for r in $(seq 1 $ROUNDS); do
    for lib in "" "$VAR"; do
        env ${lib:+ESR_LIB_PATH=$ROOT/$lib} python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other --no-optimizer "$@" 2>/dev/null | tail -1 | \
            python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step_instrumented']; print('round $r ${lib:-in-tree}:', round(d['ms_per_step'],4), {n: round(v,4) for n, v in sorted(k.items(), key=lambda kv: -kv[1]) if v > 0.004})"
    done
done
