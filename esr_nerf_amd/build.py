"""Build libesr_hip.so (in-tree) with hipcc for gfx950.

``python -m esr_nerf_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the built library travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libesr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# NO packed-fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 / v_pk_mov_b32) in any product kernel: the target feature is
# switched off (`-target-feature -packed-fp32-ops`): whatever builds <2 x float> operations -- the SLP vectoriser, the loop
# vectoriser, a sum of two f32x4 accumulators in the source -- is scalarised by the backend.
# Why (round 6, DESIGN 5 "packed fp32"): on this hardware a packed-fp32 instruction whose op_sel reads a HIGH half for the low result
# (v_pk_add_f32 / v_pk_mul_f32 ... op_sel:[0,1]) returns wrong results in lanes 48-63 while ANOTHER wave of the same SIMD alternates
# MFMAs with op_sel'd v_fma_mix_f32 -- which is what the split-fp16 kernels do.  tools/ubench/pk_beside_mfma.hip shows it with nothing
# of this library involved (6.9 M wrong results in 3000 launches, none in any other lane quarter, none beside any other load);
# esr_expgrad_fwd, whose x / y interpolation weights the SLP vectoriser had packed exactly so, returned wrong rows in 44 % of its
# launches beside the C2 step and in 1.5 % of the light-transport steps of two ranks sharing a card (profiles/r06_packed_fp32_lanes.txt).
# Cost: none -- three builds alternating on one box, C2 step: packed 2.066-2.083 ms, this build 2.053-2.070 ms, this build without
# the SLP vectoriser as well 2.091-2.110 ms (the vectoriser's merged loads and stores are worth keeping; C4 / C3 bf16: within 1 %
# of each other: profiles/r06_ab_packed_fp32_builds.txt).  tests/test_isa.py asserts that no kernel of the library holds a
# packed-fp32 instruction.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-fno-fast-math", *NO_PACKED_FP32]
FLAGS += os.environ.get("ESR_EXTRA_HIPCC_FLAGS", "").split()      # developer experiments (-DESR_EXP_...), build time only


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(paths):
    h = hashlib.sha1()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA.items())).encode())
    return h.hexdigest()


# per-source flags.
# mlp_split.hip, which runs ONE wave per SIMD on ~430 registers: `-mllvm -amdgpu-mfma-vgpr-form=1` keeps the MFMA accumulators in
# the VGPR half -- the epilogue's vector instructions read them there instead of through v_accvgpr_read: 600 -> 367 accumulation-
# register moves per tile group in the radiance forward (5441 -> 5198 instructions), 466 -> 234 in the input gradients (4556 ->
# 4321).  Round 4 measured the flag slower on that round's kernel (three accumulator sets: the planes' traffic moved to
# v_accvgpr_write, 0.705 -> 0.72 ms); on today's kernels, three alternating rounds on one box: radiance forward 0.491-0.495 ->
# 0.479-0.483 ms, C2 step 1.981-1.994 -> 1.972-1.983 ms (round 6; the same flag on mlp.hip moves nothing at C3 / C4 / C5).
# tone_wgrad.hip: the SLP vectoriser pairs the per-lane fp32 sums of the weight-gradient kernels up in register PAIRS -- 150 register
# moves per tile and 100+ more live registers in tone_wgrad_split_t_kernel (367 registers, or 320 bytes of scratch at two waves per
# SIMD; 194 registers and no moves without it: round 6).
EXTRA = {"tone_wgrad.hip": ["-fno-slp-vectorize"], "mlp_split.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "esr_hip.h"))
    stamp = _stamp(deps)
    sfile = obj + ".stamp"
    if os.path.exists(obj) and os.path.exists(sfile) and open(sfile).read() == stamp:
        return obj, False
    cmd = [HIPCC, *FLAGS, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    # (the host pass of the same command does not know the device feature: "'-packed-fp32-ops' is not a recognized feature for
    #  this target (ignoring feature)", once per function -- dropped; everything else the compiler says is passed on)
    err = "\n".join(l for l in r.stderr.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in l)
    if err.strip():
        sys.stderr.write(err + "\n")
    with open(sfile, "w") as f:
        f.write(stamp)
    return obj, True


def build_lib(force: bool = False, jobs: int = 4) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
