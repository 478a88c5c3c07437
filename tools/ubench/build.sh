#!/bin/bash
# Build the micro-benchmarks for gfx950 (hipcc cross-compiles without a GPU); run them with
#   gpurun -- './tools/ubench/lds_atomic'   etc.
cd "$(dirname "$0")"
for f in lds_atomic lds_dma_m0 mfma_f32_loop mfma_vmem_mix mfma4_loop mfma_mix permlane_swap; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $f $f.hip 2>&1 | grep -E "error" 
done
ls -la
# the stamp harness includes the product kernel source
for f in fwd_stamps dgrad_stamps fwd16_stamps tone_stamps; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -I../../include -I../../esr_nerf_amd/csrc -o $f $f.hip 2>&1 | grep -E "error"
done
