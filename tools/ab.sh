#!/bin/bash
# A/B of environment toggles on ONE box (box-to-box spread is ~2 %): tools/ab.sh "VAR=0" "VAR=1" [bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    v=$(env $cfg python bench.py --no-cpu-baseline --no-optimizer --no-kernel-timing --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$cfg $v"
  done
done
