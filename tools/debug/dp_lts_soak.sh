#!/bin/bash
# N runs of tests/test_gpu_dp.py's two-rank light-transport worker (two ranks on ONE card over gloo), then tools/debug/lts_repeat.py in two
# processes side by side: the soak behind "0 failures" in profiles/r06_packed_fp32_lanes.txt.
#   gpurun --timeout 1200 -- 'bash tools/debug/dp_lts_soak.sh 150'
N=${1:-60}
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import re
src = open("tests/test_gpu_dp.py").read()
open("/tmp/lts_worker.py", "w").write(re.search(r"LTS_WORKER = r'''(.*?)'''", src, re.S).group(1))
PY
fail=0
for i in $(seq 1 $N); do
  MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 /tmp/lts_worker.py $PWD > /tmp/dp_run.txt 2>&1 || { fail=$((fail+1)); grep "AssertionError" /tmp/dp_run.txt | head -2; }
  [ $((i % 25)) = 0 ] && echo "   $i runs, $fail failed"
done
echo "two-rank light-transport worker: $fail failures of $N runs"
export ESR_REPEAT_SEEDED=1
python tools/debug/lts_repeat.py default 120 > /tmp/rep_a.txt 2>&1 &
python tools/debug/lts_repeat.py default 120 > /tmp/rep_b.txt 2>&1
wait
grep "repetitions differ" /tmp/rep_a.txt /tmp/rep_b.txt
