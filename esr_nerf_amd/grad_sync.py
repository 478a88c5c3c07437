"""Data-parallel exchange of the dense-grid gradients.

Rays are sharded over ranks, parameters are replicated, so once per step every rank needs the SUM
of all ranks' gradients.  > 99 % of that payload is the dense grids (sdf + colour grids: 218 MB
fp32 at C2), and a ray batch touches only part of them.  ``GridGradSync.reduce`` therefore

  1. marks the 512-byte bricks of the flat gradient buffer that are non-zero on this rank
     (``esr_brick_flags``, one streaming read),
  2. forms the union over ranks (all-reduce MAX of one byte per brick: 0.4 MB at C2),
  3. packs the union's bricks (``esr_brick_pack``), all-reduces the packed buffer over RCCL / xGMI,
     and scatters the result back (``esr_brick_unpack``);
  4. falls back to one dense all-reduce of the whole buffer when the union covers more than
     ``dense_above`` of the bricks (the pack / unpack passes would cost more than they save).

Every rank takes the same branch (the decision is made on the reduced flags), so the collective
sequence is identical everywhere.  Step 2 ends in one host<->device sync (the brick count sizes the
collective); the caller enqueues all remaining compute BEFORE calling ``reduce`` so the device stays
busy while the host waits (FineStep / LtsStep run the weight gradients on a second stream).

The three brick kernels are HIP (csrc/brick.hip) and refuse CPU tensors; the world-size-2 gloo tests
inject a torch restatement of them through ``ops=`` to rehearse the protocol on CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib


class HipBrickOps:
    """csrc/brick.hip through the C ABI."""

    def __init__(self):
        self.L = _lib.lib()
        self.brick = int(self.L.esr_brick_floats())

    def flags(self, flat: torch.Tensor, out: torch.Tensor):
        _lib.check(self.L.esr_brick_flags(_lib.ptr(flat), C.c_int64(flat.numel()), _lib.ptr(out),
                                          _lib.stream_ptr(flat.device)), "esr_brick_flags")

    def pack(self, flat: torch.Tensor, idx: torch.Tensor, packed: torch.Tensor):
        _lib.check(self.L.esr_brick_pack(_lib.ptr(flat), C.c_int64(flat.numel()), _lib.ptr(idx),
                                         C.c_int64(idx.numel()), _lib.ptr(packed),
                                         _lib.stream_ptr(flat.device)), "esr_brick_pack")

    def unpack(self, packed: torch.Tensor, idx: torch.Tensor, flat: torch.Tensor):
        _lib.check(self.L.esr_brick_unpack(_lib.ptr(packed), _lib.ptr(idx), C.c_int64(idx.numel()),
                                           _lib.ptr(flat), C.c_int64(flat.numel()),
                                           _lib.stream_ptr(flat.device)), "esr_brick_unpack")


class GridGradSync:
    def __init__(self, process_group, dense_above: float = 0.8, ops=None):
        self.pg = process_group
        self.dense_above = float(dense_above)
        self.ops = ops if ops is not None else HipBrickOps()
        self.brick = self.ops.brick
        self._flags: Optional[torch.Tensor] = None
        self._packed: Optional[torch.Tensor] = None
        self.last = dict(bricks=0, sent=0, mode="none")

    def reduce(self, flat: torch.Tensor):
        """In-place sum of ``flat`` (1-D fp32, the dense-grid part of the gradient buffer) over the
        group.  Blocks the host once (brick count); stream-ordered otherwise."""
        n = flat.numel()
        if n == 0:
            return
        nb = (n + self.brick - 1) // self.brick
        if self._flags is None or self._flags.numel() != nb or self._flags.device != flat.device:
            self._flags = torch.empty(nb, dtype=torch.uint8, device=flat.device)
        self.ops.flags(flat, self._flags)
        dist.all_reduce(self._flags, op=dist.ReduceOp.MAX, group=self.pg)
        idx = self._flags.nonzero().view(-1)              # the one host sync: sizes the collective
        k = idx.numel()
        if k >= self.dense_above * nb:
            dist.all_reduce(flat, group=self.pg)
            self.last = dict(bricks=nb, sent=nb, mode="dense")
            return
        if k:
            if self._packed is None or self._packed.numel() < k * self.brick or self._packed.device != flat.device:
                self._packed = torch.empty(max(k * self.brick, nb * self.brick // 2), dtype=torch.float32,
                                           device=flat.device)
            packed = self._packed[: k * self.brick]
            self.ops.pack(flat, idx, packed)
            dist.all_reduce(packed, group=self.pg)
            self.ops.unpack(packed, idx, flat)
        self.last = dict(bricks=nb, sent=k, mode="sparse")
