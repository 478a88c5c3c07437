#!/bin/bash
# The round's evidence set on ONE build, two calls (each fits a gpurun call of 1200 s):
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh r05_z bench'   bench lines of every workload, dispatch counts, host profiles
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh r05_z prof'    kernel stats + FETCH / WRITE passes of C2 f32, C2 bf16, C3 bf16,
#                                                                         C4 lts f32 and C5 pdra bf16, the matrix-pipe counters of C2
# Summaries: tools/pmc_summary.py / mfma_summary.py (CPU side), then copy what is cited into profiles/.
TAG=${1:-final}; PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
if [ "$PART" = bench ] || [ "$PART" = all ]; then
    bash tools/bench_all.sh $TAG | tail -14
    bash tools/dispatches.sh ${TAG}_c5 --config C5 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c5_dispatches.txt
    bash tools/dispatches.sh ${TAG}_c4 --config C4 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c4_dispatches.txt
    bash tools/dispatches.sh ${TAG}_c2 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c2_dispatches.txt
    bash tools/dispatches.sh ${TAG}_c3bf16 --config C3 --dtype bf16 > /dev/null 2>&1; head -3 gpurun_out/${TAG}_c3bf16_dispatches.txt
    python3 tools/host_profile.py --config C5 --steps 20 --top 12 > gpurun_out/${TAG}_host_c5.txt 2>&1; grep "ms per step" gpurun_out/${TAG}_host_c5.txt
    python3 tools/host_profile.py --config C2 --steps 20 --top 12 > gpurun_out/${TAG}_host_c2.txt 2>&1; grep "ms per step" gpurun_out/${TAG}_host_c2.txt
    python3 tools/host_gaps.py --config C5 > gpurun_out/${TAG}_host_gaps_c5.txt 2>&1; tail -2 gpurun_out/${TAG}_host_gaps_c5.txt
    python3 tools/host_gaps.py --config C2 > gpurun_out/${TAG}_host_gaps_c2.txt 2>&1; tail -2 gpurun_out/${TAG}_host_gaps_c2.txt
fi
if [ "$PART" = prof ] || [ "$PART" = all ]; then
    bash tools/profile_workload.sh ${TAG}_c2 | tail -3
    bash tools/profile_workload.sh ${TAG}_c2bf16 --dtype bf16 | tail -3
    bash tools/profile_workload.sh ${TAG}_c3bf16 --config C3 --dtype bf16 | tail -3
    bash tools/profile_workload.sh ${TAG}_c4 --config C4 | tail -3
    bash tools/profile_workload.sh ${TAG}_c5 --config C5 | tail -3
    bash tools/profile_mfma.sh $TAG | tail -3
fi
