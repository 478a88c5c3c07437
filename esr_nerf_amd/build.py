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
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-fno-fast-math"]
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


# per-source flags (none at present).  Tried for mlp_split.hip, which runs ONE wave per SIMD on ~460 registers:
# `-mllvm -amdgpu-mfma-vgpr-form=1` keeps the MFMA accumulators in the VGPR half (1163 -> 477 v_accvgpr_read per tile group,
# 6.6 k -> 6.1 k instructions in the loop body) but moves the planes' traffic to v_accvgpr_write: 0.705 -> 0.72 ms at C2.
# tone_wgrad.hip: the SLP vectoriser packs the per-lane fp32 sums of the weight-gradient kernels into v_pk_fma_f32 / v_pk_add_f32 on
# register PAIRS -- 150 register moves per tile and 100+ more live registers in tone_wgrad_split_t_kernel (367 registers, or
# 320 bytes of scratch at two waves per SIMD; 194 registers and no moves without it: round 6).
EXTRA = {"tone_wgrad.hip": ["-fno-slp-vectorize"]}


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
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
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
