#!/bin/bash
# The round's evidence set, one call on one box:   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh r04_z'
#   bench lines of every workload, kernel stats + FETCH / WRITE passes of C2 f32, C3 bf16 and C5, the matrix-pipe counters of
#   C2, dispatch counts, host-side profile of the LTS step.  Summaries: tools/pmc_summary.py / mfma_summary.py (CPU side).
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/bench_all.sh $TAG | tail -14
bash tools/profile_workload.sh ${TAG}_c2 | tail -3
bash tools/profile_workload.sh ${TAG}_c3bf16 --config C3 --dtype bf16 | tail -3
bash tools/profile_workload.sh ${TAG}_c5 --config C5 | tail -3
bash tools/profile_mfma.sh $TAG | tail -3
bash tools/dispatches.sh ${TAG}_c5 --config C5 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c5_dispatches.txt
bash tools/dispatches.sh ${TAG}_c2 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c2_dispatches.txt
bash tools/dispatches.sh ${TAG}_c3bf16 --config C3 --dtype bf16 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c3bf16_dispatches.txt
python3 tools/host_profile.py --config C5 --steps 20 --top 12 > gpurun_out/${TAG}_host_c5.txt 2>&1; grep "ms per step" gpurun_out/${TAG}_host_c5.txt
python3 tools/host_profile.py --config C2 --steps 20 --top 12 > gpurun_out/${TAG}_host_c2.txt 2>&1; grep "ms per step" gpurun_out/${TAG}_host_c2.txt
