#!/bin/bash
# esr_expgrad_fwd (tools/debug/two_process_lanes.py: the same launch 30000 times, every result compared with the first) while ANOTHER
# process runs the C4 light-transport step on the same card: packed build of lts.hip / in-tree library / packed build again.
#   bash tools/debug/build_lts_slp_variant.sh && gpurun -- "bash tools/debug/expgrad_beside_step.sh"
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="--no-cpu-baseline --no-other --no-optimizer --no-kernel-timing --warmup 5"
python bench.py --config C4 --steps 9000 $B > gpurun_out/agg.txt 2>&1 &
sleep 12
for lib in tools/_variants/lts_slp.so "" tools/_variants/lts_slp.so; do
  echo "== ${lib:-in-tree library (no packed fp32)}"
  env ${lib:+ESR_LIB_PATH=$PWD/$lib} timeout -k 5 200 python tools/debug/two_process_lanes.py 30000 victim 2>&1 | grep "esr_expgrad_fwd:" | cut -c1-250
done
wait
tail -1 gpurun_out/agg.txt | cut -c1-160
