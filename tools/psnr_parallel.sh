#!/bin/bash
# Teacher-student PSNR statistics over a seed range, in the background with a progress line a minute (a quiet call is taken for
# hung) and a part file rewritten after every seed (a call cut at its limit keeps what it measured):
#   gpurun -- 'bash tools/psnr_parallel.sh <name> <stage> <steps> <seed0> <seed1> [PROCS] [extra tool args, e.g. --other f32mfma]'
# -> gpurun_out/psnr/<name>.json (seeds seed0 .. seed1-1, statistics over all of them) + per-process logs.  Merge several
# calls with: python tools/psnr_teacher_student.py --merge gpurun_out/psnr/<a>.json gpurun_out/psnr/<b>.json --summary ...
# PROCS (default 1): processes side by side on the one GPU.  Measured, fine stage: 1 process 2.2 seeds/s, 4 processes 1.0 seeds/s
# together -- processes time-slice the card, they do not share it -- so 1 is the right number; the argument stays for boxes
# with more than one GPU visible.
NAME=$1; STAGE=$2; STEPS=$3; S0=$4; S1=$5; PROCS=${6:-1}; shift $(( $# < 6 ? $# : 6 ))
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/psnr
mkdir -p "$OUT"
N=$((S1 - S0)); PER=$(((N + PROCS - 1) / PROCS))
pids=()
for p in $(seq 0 $((PROCS - 1))); do
    a=$((S0 + p * PER)); b=$((a + PER)); [ $b -gt $S1 ] && b=$S1
    [ $a -ge $b ] && continue
    python3 "$ROOT/tools/psnr_teacher_student.py" --stage "$STAGE" --steps "$STEPS" --seed-start $a --seeds-range $b "$@" \
        --summary "$OUT/$NAME.part$p.json" > "$OUT/$NAME.part$p.log" 2>&1 &
    pids+=($!)
done
# a progress line a minute while the processes run
while :; do
    alive=0; for pid in "${pids[@]}"; do kill -0 $pid 2>/dev/null && alive=1; done
    [ $alive -eq 0 ] && break
    sleep 60
    echo "[psnr_parallel $NAME] seeds done: $(cat "$OUT/$NAME".part*.log 2>/dev/null | grep -c '^seed ')/$N"
done
rc=0; for pid in "${pids[@]}"; do wait $pid || rc=1; done
python3 "$ROOT/tools/psnr_teacher_student.py" --merge "$OUT/$NAME".part*.json --summary "$OUT/$NAME.json" | cut -c1-500
exit $rc
