#!/bin/bash
# Build the micro-benchmarks for gfx950 (hipcc cross-compiles without a GPU); run them with
#   gpurun -- './tools/ubench/lds_atomic'   etc.
cd "$(dirname "$0")"
for f in lds_atomic lds_dma_m0 mfma_f32_loop mfma_vmem_mix mfma4_loop mfma_mix permlane_swap mfma_f32_shapes mfma_valu_overlap issue_cost asm_behind_mfma pk_beside_mfma reg_canary; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $f $f.hip 2>&1 | grep -E "error" 
done
ls -la
# the stamp harness includes the product kernel source
for f in fwd_stamps dgrad_stamps fwd16_stamps tone_stamps split_stamps ldsread_srcc; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -I../../include -I../../esr_nerf_amd/csrc -o $f $f.hip 2>&1 | grep -E "error"
done
# timing variants of the split forward's stamps (wrong results, one ingredient removed each)
for v in NO_MFMA NO_HSTORE NO_WREAD H24; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -DESR_SPLIT_$v -I../../include -I../../esr_nerf_amd/csrc \
      -o split_stamps_$(echo $v | tr 'A-Z' 'a-z') split_stamps.hip 2>&1 | grep -E "error"
done
# round 6's prototype of the radiance kernels at two waves per SIMD (nsplit_proto.h) beside the product kernels, and its timing variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -I../../include -I../../esr_nerf_amd/csrc -o nsplit_bench nsplit_bench.hip 2>&1 | grep -E "error"
for v in NO_MFMA NO_HSTORE NO_WREAD NO_STAGE; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -DESR_NS_$v -I../../include -I../../esr_nerf_amd/csrc \
      -o nsplit_bench_$(echo $v | tr 'A-Z' 'a-z') nsplit_bench.hip 2>&1 | grep -E "error"
done
