#!/bin/bash
# A/B of builds of the library inside the real step, alternating on ONE box (box-to-box spread is ~2 %):
#   gpurun -- 'bash tools/ab_lib.sh tools/_variants/X.so[,tools/_variants/Y.so] [rounds] [bench args]'      (A = the in-tree library)
VARS=$1; ROUNDS=${2:-3}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for r in $(seq 1 $ROUNDS); do
    for lib in "" ${VARS//,/ }; do
        ms=$(env ${lib:+ESR_LIB_PATH=$ROOT/$lib} python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other --no-optimizer --no-kernel-timing "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
        echo "round $r ${lib:-in-tree}: $ms ms/step"
    done
done
