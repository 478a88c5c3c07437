"""Checks on the GENERATED gfx950 code (`hipcc -S` of the MLP sources; CPU only, no GPU needed).

1. No hand-written (inline-asm) vector instruction reads a register that an MFMA wrote fewer than HAZARD wait states
   earlier.  hipcc's hazard recogniser inserts the wait states a VALU instruction needs behind the MFMA that produced its
   operand, but it does not look inside asm statements: round 3's bf16 tone-mapper kernel read an accumulator with an
   asm `v_max_f32` before its single 8-pass MFMA had written it ("compiler hazard").  The ReLU is a
   compiler-visible integer max since round 4; this test keeps every remaining / future asm statement honest.
2. The ReLU of the MLP kernels is ONE instruction per accumulator element (`v_max_i32`), not fmaxf's canonicalise + max.
3. Register budgets that performance depends on: no scratch in the pack kernel (round 3: 256 B per lane), no spilled
   vector registers in the bf16 radiance forward (round 3: 12).
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_meta as km      # noqa: E402

# wait states between an XDL (MFMA) write of a VGPR and a VALU read of it: up to 18 for a 16-pass instruction on
# gfx940-class hardware (LLVM's GCNHazardRecognizer: 2-pass 5, 4-pass 7 / 8-pass 11 / 16-pass 19 for some pairs); the
# test asks for the largest figure whatever the MFMA, which is what "safe by construction" should mean
HAZARD = 19


def _asm(name):
    return km.asm_of(os.path.join(ROOT, "esr_nerf_amd", "csrc", name))


def _regs(tok):
    """VGPR / AGPR numbers named by an operand token: v12, a3, v[4:7], a[0:15]."""
    out = set()
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(kind, i) for i in range(int(lo), int(hi) + 1)}
    for kind, n in re.findall(r"\b([va])(\d+)\b", tok):
        out.add((kind, int(n)))
    return out


def asm_reads_behind_mfma(path, every_reader_in=None, hazard=None):
    """[(kernel, line number, instruction, distance)] of inline-asm instructions whose sources an MFMA wrote < HAZARD
    wait states before (linear scan per function: conservative across branches).  ``every_reader_in``: a substring of
    kernel names in which EVERY vector / LDS-write / buffer-store instruction is held to the bound, not only the hand-written
    ones -- for kernels that pin their MFMAs with empty asm statements, behind which hipcc no longer provides the wait
    states (round 5's wave-pair kernels, since removed: a third of the tiles came out wrong when a tile's sums were read a
    few instructions behind its last pinned MFMA).  ``hazard``: the bound for those kernels (8-pass MFMAs: 11 wait states)."""
    HZ = hazard or HAZARD
    bad, recent, in_asm, func, slot = [], [], False, None, 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            func, recent, slot = m.group(1), [], 0
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        rest = rest.split(";")[0]
        if op == "s_nop":
            slot += int(rest.strip() or 0) + 1
            continue
        slot += 1
        ops = [o.strip() for o in rest.split(",")]
        if op.startswith(("v_mfma", "v_smfmac")):
            recent.append((slot, _regs(ops[0])))
            recent = [(s, r) for s, r in recent if slot - s < 4 * HAZARD]
            continue
        everyone = every_reader_in is not None and func is not None and every_reader_in in func
        if (in_asm and op.startswith("v_")) or (everyone and op.startswith(("v_", "ds_write", "buffer_store"))):
            first_src = 0 if op.startswith(("ds_write", "buffer_store")) else 1
            srcs = set().union(*[_regs(o) for o in ops[first_src:]]) if len(ops) > first_src else set()
            for s, dst in recent:
                if slot - s < (HZ if everyone else HAZARD) and srcs & dst:
                    bad.append((func, ln, t, slot - s))
        # A register that a later NON-MFMA instruction has written no longer holds the MFMA's result: its next reader reads that
        # instruction's value (vector -> vector needs no wait state, and the write itself -- if compiler-generated -- was placed
        # behind the MFMA by the compiler's own hazard recogniser).  Matters since round 6: with the accumulators in the
        # architected half of the register file (mlp_split.hip is built with -amdgpu-mfma-vgpr-form=1) an accumulator register
        # and a vector temporary can carry the same NUMBER within 19 slots.
        if not in_asm and ops and (op.startswith("v_") and not op.startswith(("v_cmp", "v_cmpx"))
                                   or op.startswith(("ds_read", "ds_load", "buffer_load", "global_load", "scratch_load"))):
            wr = _regs(ops[0])
            if wr:
                recent = [(s, dst - wr) for s, dst in recent]
    return bad


@pytest.mark.parametrize("src", ["mlp.hip", "mlp_bf16.hip", "tone_wgrad.hip", "mlp_split.hip"])
def test_no_inline_asm_reads_a_fresh_mfma_result(src):
    bad = asm_reads_behind_mfma(_asm(src))
    assert not bad, bad[:5]


def test_scanner_catches_a_planted_hazard(tmp_path):
    p = tmp_path / "k.s"
    p.write_text("_Zfoo:\n\tv_mfma_f32_32x32x2_f32 v[0:15], v16, v17, v[0:15]\n\t;;#ASMSTART\n\tv_max_f32 v3, v3, 0\n"
                 "\t;;#ASMEND\n\ts_nop 15\n\ts_nop 3\n\t;;#ASMSTART\n\tv_max_f32 v4, v4, 0\n\t;;#ASMEND\n"
                 "\tv_max_i32_e32 v5, 0, v5\n.Lfunc_end0:\n")
    bad = asm_reads_behind_mfma(str(p))
    assert len(bad) == 1 and bad[0][2].startswith("v_max_f32 v3") and bad[0][3] == 1


def test_relu_is_one_compiler_visible_instruction():
    txt = open(_asm("mlp.hip")).read()
    body = txt[txt.index("mlp_fwd_kernelILi0E"):]
    body = body[: body.index(".Lfunc_end")]
    n_imax = len(re.findall(r"\bv_max_i32", body))
    n_fmax = len(re.findall(r"\bv_max(_num)?_f32", body))
    # 3 hidden layers x 6 tiles x 16 registers = 288 ReLUs per tile pass, in two net variants of the merged launch
    assert n_imax >= 288 and n_fmax == 0, (n_imax, n_fmax)


def test_register_budgets():
    meta = km.kernel_meta(_asm("mlp.hip"))
    pack = [v for k, v in meta.items() if "pack_kernel" in k]
    assert pack and all(v.get("scratch", 0) == 0 for v in pack), pack
    meta16 = km.kernel_meta(_asm("mlp_bf16.hip"))
    fwd0 = [v for k, v in meta16.items() if "mlp_fwd16s_kernelILi0E" in k]
    assert fwd0 and fwd0[0].get("spill_v", 0) == 0 and fwd0[0].get("scratch", 0) == 0, fwd0
    # the split-fp16 kernels run one wave per SIMD on the whole register file: spills into the AGPR half are fine, scratch is not
    metas = km.kernel_meta(_asm("mlp_split.hip"))
    for name in ("mlp_fwd_split_kernel", "mlp_dgrad_split_kernel"):
        ks = [v for k, v in metas.items() if name in k]
        assert ks and all(v.get("scratch", 0) == 0 for v in ks), (name, ks)
        # ... and the 128-wide nets' / the tone mapper's instantiations fit TWO waves per SIMD (256 registers in all)
        small = [v for k, v in metas.items() if name in k and "ILi0E" not in k]
        assert len(small) == 3 and all(v.get("vgpr", 0) + v.get("agpr", 0) <= 256 for v in small), (name, small)
    twm = km.kernel_meta(_asm("tone_wgrad.hip"))
    tw = [v for k, v in twm.items() if "tone_wgrad" in k and "reduce" not in k]
    assert len(tw) == 3 and all(v.get("scratch", 0) == 0 for v in tw), tw
    # round 6: the split-fp16 twin fits TWO waves per SIMD with no accumulation-register moves (built without the SLP
    # vectoriser: esr_nerf_amd/build.py EXTRA -- with it the kernel asks for 367 registers or 316 bytes of scratch)
    sp = [v for k, v in twm.items() if "tone_wgrad_split_t_kernel" in k]
    assert len(sp) == 1 and sp[0].get("vgpr", 999) + sp[0].get("agpr", 0) <= 256 and sp[0].get("occupancy") == 2, sp
    body = open(_asm("tone_wgrad.hip")).read()
    body = body[body.index("tone_wgrad_split_t_kernel"):]
    body = body[: body.index(".Lfunc_end")]
    assert len(re.findall(r"\bv_accvgpr_", body)) == 0 and len(re.findall(r"\bv_pk_(fma|add|mul)_f32", body)) == 0


def test_split_kernels_evaluate_their_step_tables_at_compile_time():
    """The first build of mlp_split.hip looked its step table up at RUN time in the later steps (chains of scalar loads per use:
    steps 5-8 took 10 k clocks instead of 2.5 k): the loop bodies hold a handful of scalar loads and branches, not hundreds."""
    txt = open(_asm("mlp_split.hip")).read()
    for name in ("mlp_fwd_split_kernelILi0E", "mlp_dgrad_split_kernelILi0E"):
        body = txt[txt.index(name + "EEv"):]
        body = body[: body.index(".Lfunc_end")]
        n_sload, n_branch = len(re.findall(r"\bs_load_dword", body)), len(re.findall(r"\bs_cbranch", body))
        assert n_sload < 40 and n_branch < 80, (name, n_sload, n_branch)


def test_no_kernel_of_the_library_holds_a_packed_fp32_instruction():
    """Round 6 (DESIGN 5 "packed fp32", tools/ubench/pk_beside_mfma.hip): v_pk_add_f32 / v_pk_mul_f32 with an op_sel that reads a
    high half return wrong results in lanes 48-63 while another wave of the SIMD alternates MFMAs with op_sel'd v_fma_mix_f32 -- the
    split-fp16 kernels' instruction mix.  The compiler wrote such instructions into esr_expgrad_fwd by itself (SLP vectoriser); the
    library is built with the packed-fp32 target feature off (esr_nerf_amd/build.py: NO_PACKED_FP32) and every source is checked."""
    import glob
    found = {}
    for src in sorted(glob.glob(os.path.join(ROOT, "esr_nerf_amd", "csrc", "*.hip"))):
        ops = re.findall(r"^\s+(v_pk_(?:add|mul|fma)_f32|v_pk_mov_b32)\b", open(_asm(os.path.basename(src))).read(), re.M)
        if ops:
            found[os.path.basename(src)] = len(ops)
    assert not found, found
