#!/bin/bash
# tools/_variants/lts_slp.so = the in-tree library with lts.hip compiled WITHOUT build.py's NO_PACKED_FP32 (the compiler's SLP vectoriser
# then writes v_pk_add_f32 ... op_sel into expgrad_kernel): the "before" of tools/debug/expgrad_beside_step.sh and two_process_lanes.py.
#   bash tools/debug/build_lts_slp_variant.sh        (CPU box; the library must have been built: python -m esr_nerf_amd.build)
cd "$(dirname "$0")/../.."
mkdir -p tools/_variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -fno-fast-math"
/opt/rocm/bin/hipcc $F -c esr_nerf_amd/csrc/lts.hip -o /tmp/lts_slp.o || exit 1
OBJS=$(ls esr_nerf_amd/_obj/*.o | grep -v "/lts.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/_variants/lts_slp.so $OBJS /tmp/lts_slp.o && ls -la tools/_variants/lts_slp.so
