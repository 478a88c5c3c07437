"""The discrete decisions of a HIP fine-stage step -- final survivor set and ReLU branches -- read back from the engine's
workspace, in the form ``oracle.fine_path.forward_training(force=...)`` takes them: the oracle then evaluates the same
piecewise-linear function on the same piece, and HIP results compare against it with NOTHING set aside (no dropped ray, no
cell or weight row at a looser tolerance).  What makes that legitimate is asserted by ``assert_legitimate``: a decision
that differs from the oracle's own was taken ON the decision boundary -- a weight within 2e-3 relative of the survivor
threshold, a ReLU pre-activation whose FLOAT64 value is within fp32 summation noise of zero."""
import torch

KINK = 4e-6        # |float64 pre-activation| of a flipped ReLU unit: 192 fp32 fma steps on partial sums of O(1) walk ~4e-7,
                   # measured flips up to 6e-7 (round 3); twice the band the round-4 tests used to drop rays at (2e-6)


def _decode_masks(M, hid_tiles):
    """[tiles, hid_tiles // 2, 64] mask words -> bool [tiles, 32 * hid_tiles, 32]: bit (it & 1) * 16 + r of word
    [it >> 1][32 h + s] <-> row 32 it + (r & 3) + 8 (r >> 2) + 4 h of sample s (csrc/mlp_common.h: store_relu_mask)."""
    T = M.shape[0]
    Mu = (M.to(torch.int64) & 0xffffffff).view(T, hid_tiles // 2, 2, 32)            # [tile, word, h, s]
    out = torch.zeros(T, 32 * hid_tiles, 32, dtype=torch.bool)
    for it in range(hid_tiles):
        for r in range(16):
            for h in range(2):
                row = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * h
                out[:, row, :] = ((Mu[:, it >> 1, h, :] >> ((it & 1) * 16 + r)) & 1).bool()
    return out


def hip_decisions(model):
    """After a training step of ``model`` (VoxurfF on the HIP path): dict(survivors=..., relu=...) for the oracle."""
    eng, lc = model.engine, model.last_counts
    tiles_on = (lc["n_on"] + 31) // 32
    tiles = tiles_on + (lc["n_off"] + 31) // 32
    ws = eng.ws
    if tiles == 0:                                          # no survivor: nothing was decided
        empty = lambda n: torch.zeros(n, 192, dtype=torch.bool)
        return dict(survivors=torch.zeros(0, dtype=torch.long),
                    relu=lambda ray_id, step_id, on: dict(emo=[empty(int(on.sum()))] * 3, off=[empty(int((~on).sum()))] * 3,
                                                          tone=[empty(len(ray_id))]))
    ray = ws["rec_ray"][: tiles * 32].cpu().long()
    step = ws["rec_step"][: tiles * 32].cpu().long()
    live = ray >= 0
    key = ray * (1 << 20) + step
    rad = [_decode_masks(ws[n][: tiles * 3 * 64].view(tiles, 3, 64).cpu(), 6) for n in ("M0", "M1", "M2")]
    tone = _decode_masks(ws["Mt"][: tiles * 3 * 64].view(tiles, 3, 64).cpu(), 6)
    slot_keys, order = torch.sort(key[live])
    slots = live.nonzero()[:, 0][order]                     # slot of every live sample, sorted by key

    def rows_of(mask, slot):                                # [tiles, rows, 32] -> [len(slot), rows]
        return mask[slot // 32, :, slot % 32]

    def relu(ray_id, step_id, on):
        k = ray_id * (1 << 20) + step_id
        pos = torch.searchsorted(slot_keys, k)
        assert bool((slot_keys[pos.clamp(max=len(slot_keys) - 1)] == k).all()), "the oracle's samples are the HIP step's"
        sl = slots[pos]
        assert bool((sl[on] < tiles_on * 32).all()) and bool((sl[~on] >= tiles_on * 32).all())
        return dict(emo=[rows_of(mk, sl[on]) for mk in rad], off=[rows_of(mk, sl[~on]) for mk in rad],
                    tone=[rows_of(tone, sl)])
    return dict(survivors=key[live], relu=relu)


def assert_legitimate(keep, flip_log, thres=1e-4, what=""):
    """Every decision the oracle took over from the HIP step and would have taken differently sits on its boundary."""
    n_thr = 0
    for name in ("threshold_flips", "alpha_flips", "sec_threshold_flips", "sec_alpha_flips"):
        tf = keep.get(name)
        if tf is not None and tf.numel():
            n_thr += int(tf.numel())
            assert float(((tf - thres).abs() / thres).max()) < 2e-3, (what, name, tf.tolist()[:8])
    n_flip = sum(n for _, n, _ in flip_log)
    worst = max([w for _, _, w in flip_log] + [0.0])
    print(f"[arbiter{' ' + what if what else ''}] survivor-threshold samples taken over: {n_thr}; ReLU branches taken over: {n_flip} "
          f"(largest |float64 pre-activation| among them {worst:.2e}, bound {KINK:g})")
    assert worst < KINK, (what, [(k, n, w) for k, n, w in flip_log if w >= KINK])
    return n_thr, n_flip


# ---- the light-transport steps (esr_nerf_amd/lts_engine.py: four sampling passes) ------------------------------------
def _pass_masks(P, net, n_hidden, hid_tiles):
    T = P.tiles_all
    words = hid_tiles // 2
    return [_decode_masks(P.bufs[f"{net}.M{l}"][: T * words * 64].view(T, words, 64).cpu(), hid_tiles) for l in range(n_hidden)]


def _keyed(P):
    """(sorted keys, slots) of a record-sampled pass: ray * 2**20 + step of every live slot."""
    T = P.tiles_all
    ray = P.bufs["rec_ray"][: T * 32].cpu().long()
    step = P.bufs["rec_step"][: T * 32].cpu().long()
    live = ray >= 0
    key = ray * (1 << 20) + step
    sk, order = torch.sort(key[live])
    return key[live], sk, live.nonzero()[:, 0][order]


def _rows(mask, slot):
    return mask[slot // 32, :, slot % 32]


def hip_decisions_lts(model):
    """After an LTS / PDRA step of ``model`` (ESRNeRF on the HIP path): the ``force`` dict of oracle.lts_path.forward_training."""
    eng = model.engine
    P0, P1, P2, P3 = eng.prim, eng.pts, eng.sec, eng.epsp
    out = {}
    keys0, sk0, slots0 = _keyed(P0)
    m0 = dict(off=_pass_masks(P0, "off", 3, 6), emo=_pass_masks(P0, "emo", 3, 6), tone=_pass_masks(P0, "tone", 1, 6),
              brdf=_pass_masks(P0, "brdf", 3, 4), emit=_pass_masks(P0, "emit", 3, 4))
    ton = P0.tiles_on

    def prim(ray_id, step_id, on):
        k = ray_id * (1 << 20) + step_id
        pos = torch.searchsorted(sk0, k)
        assert bool((sk0[pos.clamp(max=len(sk0) - 1)] == k).all()), "the oracle's primary samples are the HIP step's"
        sl = slots0[pos]
        assert bool((sl[on] < ton * 32).all()) and bool((sl[~on] >= ton * 32).all())
        return dict(emo=[_rows(mk, sl[on]) for mk in m0["emo"]], off=[_rows(mk, sl) for mk in m0["off"]],
                    tone=[_rows(mk, sl) for mk in m0["tone"]], brdf=[_rows(mk, sl) for mk in m0["brdf"]],
                    emit=[_rows(mk, sl) for mk in m0["emit"]])
    out["prim_survivors"], out["prim"] = keys0, prim
    # the points' pass: slot k = row k of the 2 P (camera | random direction) rows
    n_pts = 2 * int(getattr(eng, "last_point_idx").numel())
    sl1 = torch.arange(n_pts)
    out["pts"] = dict(off=[_rows(mk, sl1) for mk in _pass_masks(P1, "off", 3, 6)],
                      emo=[_rows(mk, sl1) for mk in _pass_masks(P1, "emo", 3, 6)])
    if P2.tiles_all:
        keys2, sk2, slots2 = _keyed(P2)
        m2 = dict(off=_pass_masks(P2, "off", 3, 6), emo=_pass_masks(P2, "emo", 3, 6))

        def sec(ray_id, step_id):
            k = ray_id * (1 << 20) + step_id
            pos = torch.searchsorted(sk2, k)
            assert bool((sk2[pos.clamp(max=len(sk2) - 1)] == k).all()), "the oracle's secondary samples are the HIP step's"
            sl = slots2[pos]
            return dict(off=[_rows(mk, sl) for mk in m2["off"]], emo=[_rows(mk, sl) for mk in m2["emo"]])
        out["sec_survivors"], out["sec"] = keys2, sec
    # the perturbed heads' pass (masks exist when the step kept them: eps_grads): slot k = reference-order sample k
    if "emit.M0" in P3.bufs and P3.tiles_all:
        m3 = int(P0.counts["m3"])
        sl3 = torch.arange(m3)
        out["eps"] = dict(emit=[_rows(mk, sl3) for mk in _pass_masks(P3, "emit", 3, 4)],
                          brdf=[_rows(mk, sl3) for mk in _pass_masks(P3, "brdf", 3, 4)])
    return out
