"""Ray-batch samplers of the trainers: ``BatchSampler`` (coarse / fine / lts stages,
reference utils2/utils.py:39-119) and ``RayGroupManager`` (pdra's uncertain / certain ray groups,
utils2/utils.py:122-303), with the reference's constructor arguments, attributes and sampling order, made
rank-aware for ray-sharded data parallelism.

MI355X-side differences, none of them visible in what ``sample()`` returns:

* The reference physically permutes EVERY ray array at each shuffle (``data[k] = data[k][b_ids]``, a full pass
  over the dataset per key and epoch) and slices batches out of the permuted copies.  Here the arrays stay
  where they were loaded and only the index vector is permuted; a batch is one row gather per key
  (``data[k][data_idxs[b_st:b_en]]``).  The rows are the same rows: the reference's permuted array IS
  ``original[data_idxs]`` at every point in time (``data_idxs`` goes through the same indexing, :83-90).
* ``rank`` / ``world``: every rank draws the same permutation (same seed, same generator state) and takes
  the ``rank``-th contiguous ``1/world`` share of each global batch (``trainer.shard_batch`` layout), so the
  union over ranks is exactly the single-process batch.  ``batch_size`` stays the GLOBAL batch size.

``data_preload: cpu`` keeps the arrays in pinned host memory and ships only the gathered batch.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


def _preload_to_cpu(cfg) -> bool:
    mode = cfg.system.data_preload
    if not (mode == "cpu" or "gpu" in mode or "cuda" in mode):
        raise AssertionError("cfg.system.data_preload must be 'cpu' or name a gpu/cuda device")
    return mode == "cpu"


def _pin(t: torch.Tensor) -> torch.Tensor:
    t = t.cpu().contiguous()
    return t.pin_memory() if torch.cuda.is_available() else t


def _share(lo: int, hi: int, rank: int, world: int):
    """This rank's contiguous share of the global batch [lo, hi) (last rank takes the remainder)."""
    per = (hi - lo) // world
    a = lo + rank * per
    return a, (hi if rank == world - 1 else a + per)


class BatchSampler:
    def __init__(self, cfg, data: Dict[str, torch.Tensor], keys: List[str], batch_size: int, batch_st: int = 0,
                 data_idxs: Optional[torch.Tensor] = None, rank: int = 0, world: int = 1):
        self.cfg = cfg
        self.device = cfg.system.device
        self.data_preload_to_cpu = _preload_to_cpu(cfg)
        self.home = "cpu" if self.data_preload_to_cpu else self.device
        self.keys = keys
        self.batch_size = batch_size
        self.batch_st = batch_st
        self.rank, self.world = rank, world
        n = len(data[keys[0]])
        idx = torch.arange(n) if data_idxs is None else data_idxs
        self.data_idxs = (_pin(idx) if self.data_preload_to_cpu else idx.to(self.device)).contiguous()
        self.data = {k: (_pin(data[k]) if self.data_preload_to_cpu else data[k].to(self.device).contiguous())
                     for k in keys}
        self.data_num = len(self.data_idxs)

    def shuffle(self):
        # the same draw as the reference (:81 / :86): a permutation of the CURRENT order
        b_ids = torch.randperm(self.data_num, device=self.home)
        self.data_idxs = self.data_idxs[b_ids].contiguous()
        self.batch_st = 0

    def filter(self, mask: torch.Tensor):
        """Keep the rays whose flag is set; ``mask`` is aligned with the current order (:93-105)."""
        self.data_idxs = self.data_idxs[mask.to(self.home)].contiguous()
        self.data_num = len(self.data_idxs)

    def current(self, key: str) -> torch.Tensor:
        """The reference's ``data[key]`` (rows in the current order)."""
        return self.data[key][self.data_idxs]

    def sample(self) -> Dict[str, torch.Tensor]:
        b_en = self.batch_st + self.batch_size
        if b_en > self.data_num:
            self.shuffle()
            b_en = self.batch_size
        b_st = self.batch_st
        self.batch_st = b_en
        lo, hi = _share(b_st, b_en, self.rank, self.world)
        rows = self.data_idxs[lo:hi]
        return {k: self.data[k][rows].to(self.device, non_blocking=True) for k in self.keys}


class RayGroupManager:
    def __init__(self, cfg, data: Dict[str, torch.Tensor], keys: List[str], uncert_batch_size: int,
                 cert_batch_size: int, uncert_batch_st: int = 0, cert_batch_st: int = 0,
                 uncert_data_idxs: Optional[torch.Tensor] = None, cert_data_idxs: Optional[torch.Tensor] = None,
                 rank: int = 0, world: int = 1):
        self.cfg = cfg
        self.device = cfg.system.device
        self.data_preload_to_cpu = _preload_to_cpu(cfg)
        self.home = "cpu" if self.data_preload_to_cpu else self.device
        self.keys = keys
        self.uncert_batch_size, self.cert_batch_size = uncert_batch_size, cert_batch_size
        self.uncert_batch_st, self.cert_batch_st = uncert_batch_st, cert_batch_st
        self.rank, self.world = rank, world
        n = len(data[keys[0]])
        u = torch.arange(n) if uncert_data_idxs is None else uncert_data_idxs
        c = torch.arange(0) if cert_data_idxs is None else cert_data_idxs
        place = (lambda t: _pin(t)) if self.data_preload_to_cpu else (lambda t: t.to(self.device).contiguous())
        self.uncert_data_idxs, self.cert_data_idxs = place(u), place(c)
        self.data = {k: place(data[k]) for k in keys}

    @property
    def uncert_data_num(self) -> int:
        return len(self.uncert_data_idxs)

    @property
    def cert_data_num(self) -> int:
        return len(self.cert_data_idxs)

    def shuffle(self):
        self.shuffle_uncert()
        self.shuffle_cert()

    def shuffle_uncert(self):
        b_ids = torch.randperm(self.uncert_data_num, device=self.home)
        self.uncert_data_idxs = self.uncert_data_idxs[b_ids].contiguous()
        self.uncert_batch_st = 0

    def shuffle_cert(self):
        b_ids = torch.randperm(self.cert_data_num, device=self.home)
        self.cert_data_idxs = self.cert_data_idxs[b_ids].contiguous()
        self.cert_batch_st = 0

    def filter(self, mask: torch.Tensor):
        """``mask`` (aligned with the uncertain group's current order): True stays uncertain, False moves to the END
        of the certain group (:251-283)."""
        mask = mask.to(self.home)
        self.cert_data_idxs = torch.cat([self.cert_data_idxs, self.uncert_data_idxs[~mask]]).contiguous()
        self.uncert_data_idxs = self.uncert_data_idxs[mask].contiguous()

    def uncert(self, key: str) -> torch.Tensor:
        return self.data[key][self.uncert_data_idxs]

    def cert(self, key: str) -> torch.Tensor:
        return self.data[key][self.cert_data_idxs]

    def sample(self) -> Dict[str, torch.Tensor]:
        u_en = self.uncert_batch_st + self.uncert_batch_size
        c_en = self.cert_batch_st + self.cert_batch_size
        if u_en > self.uncert_data_num:
            self.shuffle_uncert()
            u_en = min(self.uncert_data_num, self.uncert_batch_size)
        if c_en > self.cert_data_num:
            self.shuffle_cert()
            c_en = min(self.cert_data_num, self.cert_batch_size)
        u_st, c_st = self.uncert_batch_st, self.cert_batch_st
        self.uncert_batch_st, self.cert_batch_st = u_en, c_en
        # each group is sharded on its own, so every rank keeps the reference's uncertain : certain mix
        ul, uh = _share(u_st, u_en, self.rank, self.world)
        cl, ch = _share(c_st, c_en, self.rank, self.world)
        rows = torch.cat([self.uncert_data_idxs[ul:uh], self.cert_data_idxs[cl:ch]])
        batch = {k: self.data[k][rows].to(self.device, non_blocking=True) for k in self.keys}
        um = torch.ones(rows.numel(), dtype=torch.bool, device=self.device)
        if c_en == c_st:
            um[:] = False          # reference quirk (:300): `mask[-0:] = False` clears the WHOLE mask when the
        elif ch > cl:              # certain group contributes no ray -- reproduced, the trainers see the same flags
            um[-(ch - cl):] = False
        batch["uncert_masks"] = um
        return batch

    def stats(self) -> Dict[str, int]:
        return dict(uncertain=self.uncert_data_num, certain=self.cert_data_num,
                    total=self.uncert_data_num + self.cert_data_num)
