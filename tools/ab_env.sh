#!/bin/bash
# A/B timing of env-selected variants of the step on ONE box (box-to-box spread is ~2 %, larger than most single changes):
#   gpurun -- 'bash tools/ab_env.sh <tag> "VAR=a VAR2=b" "VAR=c" ... [-- bench args]'
# every variant: bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-optimizer; prints ms/step and the kernel breakdown
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
VARS=(); ARGS=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; ARGS=("$@"); break; fi; VARS+=("$1"); shift; done
i=0
for v in "${VARS[@]}"; do
  out=gpurun_out/${TAG}_ab_$i.json
  env $v python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-optimizer "${ARGS[@]}" > $out 2> gpurun_out/${TAG}_ab_$i.err || { echo "variant '$v' failed"; tail -5 gpurun_out/${TAG}_ab_$i.err; }
  python3 - "$v" $out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = (d.get("kernel_ms_per_step_instrumented") or d.get("kernel_ms_per_step_warmup", {}))
    top = ", ".join(f"{a} {b:.3f}" for a, b in list(k.items())[:12])
    print(f"[{sys.argv[1]}] ms/step {d['ms_per_step']:.4f}  rays/s {d['value']:.0f}  loss {d.get('loss')}\n    {top}")
except Exception as e:
    print(f"[{sys.argv[1]}] no result: {e}")
PY
  i=$((i+1))
done
