"""Registers, scratch, LDS and spills per kernel from `hipcc -S` of csrc/*.hip (or the files named on the command line):
   python tools/kernel_meta.py [mlp_bf16.hip ...]      (CPU only; tests/test_isa.py asserts on it)"""
import glob, os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-fno-fast-math",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",        # = build.py: NO_PACKED_FP32
         "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only"]


def _build_extra():
    """The per-source flags of the product build (esr_nerf_amd/build.py: EXTRA), read from that file so the two cannot drift."""
    src = open(os.path.join(ROOT, "esr_nerf_amd", "build.py")).read()
    m = re.search(r"^EXTRA\s*=\s*(\{.*?\})\s*$", src, re.M | re.S)
    import ast
    return {k: tuple(v) for k, v in ast.literal_eval(m.group(1)).items()} if m else {}


EXTRA = _build_extra()     # per-source flags, as esr_nerf_amd/build.py


def asm_of(src, out=None, extra=()):
    extra = tuple(extra) + EXTRA.get(os.path.basename(src), ())
    out = out or os.path.join("/tmp", "esr_asm_" + os.path.basename(src) + ".s")
    stamp = out + ".stamp"
    deps = [src] + glob.glob(os.path.join(os.path.dirname(src), "*.h")) + [os.path.join(ROOT, "include", "esr_hip.h")]
    key = str([(d, os.path.getmtime(d)) for d in deps]) + str(extra) + str(FLAGS)
    if not (os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == key):
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        open(stamp, "w").write(key)
    return out


def kernel_meta(asm_path):
    """{mangled name: dict(vgpr, agpr, sgpr, scratch, lds, spill_v, spill_s, occupancy)} from the assembler comments."""
    out, cur = {}, None
    keys = {"NumVgprs": "vgpr", "NumAgprs": "agpr", "NumSgprs": "sgpr", "ScratchSize": "scratch", "LDSByteSize": "lds",
            "Occupancy": "occupancy"}
    for line in open(asm_path):
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"^;\s*(\w+):\s*(\d+)", line)
        if m and cur is not None and m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
        m = re.match(r"^\s*\.(sgpr|vgpr)_spill_count:\s*(\d+)", line)
        if m:
            pass
    # spills live in the metadata yaml at the end: .name / .vgpr_spill_count / .sgpr_spill_count
    name = None
    for line in open(asm_path):
        m = re.match(r"^\s*\.name:\s*(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"^\s*\.(sgpr|vgpr)_spill_count:\s*(\d+)", line)
        if m and name in out:
            out[name]["spill_" + m.group(1)[0]] = int(m.group(2))
        m = re.match(r"^\s*\.private_segment_fixed_size:\s*(\d+)", line)
        if m and name in out:
            out[name]["scratch"] = int(m.group(1))
    return out


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "esr_nerf_amd", "csrc", "*.hip")))
    for f in files:
        meta = kernel_meta(asm_of(os.path.join(ROOT, "esr_nerf_amd", "csrc", f)))
        for n, d in meta.items():
            print(f"{f:16s} v {d.get('vgpr', 0):3d} a {d.get('agpr', 0):3d} s {d.get('sgpr', 0):3d} scratch {d.get('scratch', 0):5d} "
                  f"spill v/s {d.get('spill_v', 0):3d}/{d.get('spill_s', 0):3d} lds {d.get('lds', 0):6d} occ {d.get('occupancy', 0)}  {demangle(n)[:90]}")
