"""Data-parallel exchange of the dense-grid gradients.

Rays are sharded over ranks, parameters are replicated, so once per step every rank needs the SUM
of all ranks' gradients.  > 99 % of that payload is the dense grids (sdf + colour grids: 218 MB
fp32 at C2), and a ray batch touches only part of them.  Two forms live here:

``GridGradSync`` (option 1 of SURVEY 8(e): dirty bricks only)
  1. marks the 512-byte bricks of the flat gradient buffer that are non-zero on this rank
     (``esr_brick_flags``, one streaming read),
  2. forms the union over ranks (all-reduce MAX of one byte per brick: 0.4 MB at C2),
  3. packs the union's bricks (``esr_brick_pack``), all-reduces the packed buffer over RCCL / xGMI,
     and scatters the result back (``esr_brick_unpack``);
  4. falls back to one dense all-reduce of the whole buffer when the union covers more than
     ``dense_above`` of the bricks (the pack / unpack passes would cost more than they save).
  NO host wait inside the exchange: the brick list has a host-known CAPACITY (the previous step's count + 25 %), it
  is built on the device (prefix sum of the flags, unused slots = -1), and the collective always moves ``capacity``
  bricks.  This step's true count travels to pinned host memory beside the exchange; ``verify()``, called after
  everything of the step has been enqueued, reads it (by then it has long landed), adapts the capacity and -- only
  if the union outgrew it -- sends the overflowing bricks in a second, exactly sized pass.  Every rank sees the same
  flags and the same capacity, so every rank takes the same branch and the collective sequence is identical.

``ShardedGrids`` (option 2: reduce-scatter + owner-shard Adam + all-gather)
  The three grids live in ONE flat parameter buffer (the module parameters are views of it, the colour grids keep
  their channels-last order), cut into G equal shards.  ``reduce_scatter`` leaves every rank with the summed
  gradient of ITS shard only (each GPU receives 1/G of the buffer from each peer: all 7 xGMI links carry traffic
  at once), the fused Adam runs on that shard (moments are 1/G of the dense optimizer's), and ``all_gather``
  returns the updated parameters.  Used by optimizer.ShardedGridAdam behind ESR_GRAD_SYNC=shard.

The brick kernels are HIP (csrc/brick.hip) and refuse CPU tensors; the gloo tests inject a torch restatement of them
through ``ops=`` to rehearse the protocols on CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

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

    def list(self, flags: torch.Tensor, cap: int):
        """(idx [cap] int64: the first ``cap`` flagged bricks ascending, unused slots -1; count [1] int64 on the device)
        in ONE call (esr_brick_list) -- the torch form is six launches over all bricks inside the step."""
        dev = flags.device
        if getattr(self, "_scratch", None) is None or self._scratch.device != dev:
            self._scratch = torch.empty(int(self.L.esr_brick_list_scratch_ints()), dtype=torch.int32, device=dev)
        idx = torch.empty(cap, dtype=torch.int64, device=dev)
        count = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.check(self.L.esr_brick_list(_lib.ptr(flags), C.c_int64(flags.numel()), C.c_int64(cap), _lib.ptr(idx),
                                         _lib.ptr(count), _lib.ptr(self._scratch), _lib.stream_ptr(dev)), "esr_brick_list")
        return idx, count

    def unpack(self, packed: torch.Tensor, idx: torch.Tensor, flat: torch.Tensor):
        _lib.check(self.L.esr_brick_unpack(_lib.ptr(packed), _lib.ptr(idx), C.c_int64(idx.numel()),
                                           _lib.ptr(flat), C.c_int64(flat.numel()),
                                           _lib.stream_ptr(flat.device)), "esr_brick_unpack")


class GridGradSync:
    def __init__(self, process_group, dense_above: float = 0.8, ops=None, headroom: float = 1.25,
                 min_capacity: int = 64):
        self.pg = process_group
        self.dense_above = float(dense_above)
        self.headroom = float(headroom)
        self.min_capacity = int(min_capacity)
        self.ops = ops if ops is not None else HipBrickOps()
        self.brick = self.ops.brick
        self._flags: Optional[torch.Tensor] = None
        self._packed: Optional[torch.Tensor] = None
        self._cap: Optional[int] = None           # capacity (bricks) of this step's list; None: learn it first
        self._pending = None                      # (flat, count_host, event, cap) of the open step
        self.last = dict(bricks=0, sent=0, mode="none")

    # ------------------------------------------------------------------ helpers
    def _ensure(self, nb: int, k: int, device):
        if self._flags is None or self._flags.numel() != nb or self._flags.device != device:
            self._flags = torch.empty(nb, dtype=torch.uint8, device=device)
        if self._packed is None or self._packed.numel() < k * self.brick or self._packed.device != device:
            self._packed = torch.empty(k * self.brick, dtype=torch.float32, device=device)

    def _exchange(self, flat: torch.Tensor, idx: torch.Tensor):
        """pack -> all-reduce -> unpack of the listed bricks (negative entries: unused slots)."""
        k = idx.numel()
        if k == 0:
            return
        self._ensure(self._flags.numel(), k, flat.device)
        packed = self._packed[: k * self.brick]
        self.ops.pack(flat, idx, packed)
        dist.all_reduce(packed, group=self.pg)
        self.ops.unpack(packed, idx, flat)

    # ------------------------------------------------------------------ the exchange
    def reduce(self, flat: torch.Tensor):
        """In-place sum of ``flat`` (1-D fp32, the dense-grid part of the gradient buffer) over the group,
        stream-ordered, without a host wait.  Call ``verify()`` once the rest of the step has been enqueued."""
        if self._pending is not None:
            self.verify()
        n = flat.numel()
        if n == 0:
            return
        nb = (n + self.brick - 1) // self.brick
        self._ensure(nb, 0, flat.device)
        self.ops.flags(flat, self._flags)
        dist.all_reduce(self._flags, op=dist.ReduceOp.MAX, group=self.pg)
        if self._cap is None:
            # first step: nothing to size the list with -- one exact (host-synchronising) pass
            idx = self._flags.nonzero().view(-1)
            k = idx.numel()
            self._cap = max(self.min_capacity, int(k * self.headroom))
            if k >= self.dense_above * nb:
                dist.all_reduce(flat, group=self.pg)
                self.last = dict(bricks=nb, sent=nb, union=k, mode="dense")
            else:
                self._exchange(flat, idx)
                self.last = dict(bricks=nb, sent=k, union=k, mode="sparse")
            return
        cap = min(self._cap, nb)
        dense = cap >= self.dense_above * nb
        count_host = torch.empty(1, dtype=torch.int64, pin_memory=flat.is_cuda)
        if hasattr(self.ops, "list"):
            # list of capacity `cap` and the union's count built by ONE device call (csrc/brick.hip: esr_brick_list)
            idx, count_dev = self.ops.list(self._flags, 0 if dense else cap)
            count_host.copy_(count_dev, non_blocking=True)
        else:
            # torch form (the CPU test double): slot r <- the r-th flagged brick; bricks of rank >= cap (the union
            # outgrew the capacity) and unflagged bricks all fall into a dump slot; unused slots stay -1
            on = self._flags.to(torch.int64)
            rank_of = torch.cumsum(on, 0) - 1                   # position of a flagged brick in the union's list
            count_host.copy_(rank_of[-1:] + 1, non_blocking=True)
            if not dense:
                slot = torch.where((on > 0) & (rank_of < cap), rank_of, torch.full_like(rank_of, cap))
                idx = torch.full((cap + 1,), -1, dtype=torch.int64, device=flat.device)
                idx.scatter_(0, slot, torch.arange(nb, dtype=torch.int64, device=flat.device))
                idx = idx[:cap]
        ev = None
        if flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        if dense:
            dist.all_reduce(flat, group=self.pg)
            self._pending = (None, count_host, ev, nb)
            self.last = dict(bricks=nb, sent=nb, mode="dense")
            return
        self._exchange(flat, idx)
        self._pending = (flat, count_host, ev, cap)
        self.last = dict(bricks=nb, sent=cap, mode="sparse")

    @torch.no_grad()
    def profile(self, flat: torch.Tensor, reps: int = 5):
        """Milliseconds of every phase of one sparse exchange of ``flat`` (HIP events on the current stream, mean of
        ``reps``): what the data-parallel step pays before / beside the wire time.  With a one-rank group the two
        collectives are local copies, so these are the fixed costs the driver's first multi-GPU run can be read against.
        Works on a copy: ``flat`` is left as it is."""
        flat = flat.clone()
        n = flat.numel()
        nb = (n + self.brick - 1) // self.brick
        self._ensure(nb, 0, flat.device)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        phases = ["flags", "allreduce_flags", "list", "pack", "allreduce_packed", "unpack"]
        acc = {k: 0.0 for k in phases}
        world = dist.get_world_size(self.pg)
        k_used = 0
        cap_known = self._cap
        if cap_known is None:           # (a sync object that has not run a step yet: size the list as its second step would)
            self.ops.flags(flat, self._flags)
            dist.all_reduce(self._flags, op=dist.ReduceOp.MAX, group=self.pg)
            cap_known = max(self.min_capacity, int(int(self._flags.sum(dtype=torch.int64)) * self.headroom))
        for _ in range(reps):
            marks = [ev() for _ in range(len(phases) + 1)]
            marks[0].record()
            self.ops.flags(flat, self._flags); marks[1].record()
            dist.all_reduce(self._flags, op=dist.ReduceOp.MAX, group=self.pg); marks[2].record()
            cap = int(cap_known)
            if hasattr(self.ops, "list"):
                idx, _ = self.ops.list(self._flags, min(cap, nb))
            else:
                idx = self._flags.nonzero().view(-1)
            marks[3].record()
            k_used = idx.numel()
            self._ensure(nb, k_used, flat.device)
            packed = self._packed[: k_used * self.brick]
            self.ops.pack(flat, idx, packed); marks[4].record()
            dist.all_reduce(packed, group=self.pg); marks[5].record()
            self.ops.unpack(packed, idx, flat); marks[6].record()
            torch.cuda.synchronize(flat.device)
            for i, k in enumerate(phases):
                acc[k] += marks[i].elapsed_time(marks[i + 1])
        out = {k: v / reps for k, v in acc.items()}
        out.update(bricks=nb, list_slots=k_used, packed_mb=round(k_used * self.brick * 4 / 1e6, 1), ranks=world)
        return out

    def verify(self):
        """Close the step's exchange: read the union's true brick count (a host wait on a copy that was enqueued
        before the packed all-reduce, i.e. long done), adapt the capacity, and send any overflow exactly."""
        if self._pending is None:
            return
        flat, count_host, ev, cap = self._pending
        self._pending = None
        if ev is not None:
            ev.synchronize()
        k = int(count_host[0])
        self.last["union"] = k
        if flat is not None and k > cap:
            # rare (the union grew by > 25 %): the flagged bricks beyond the list's capacity, from the flags, which stay
            # intact until the next reduce()
            over = (self._flags > 0).nonzero().view(-1)[cap:]
            self._exchange(flat, over)
            self.last["sent"] += over.numel()
            self.last["overflow"] = over.numel()
        self._cap = max(self.min_capacity, int(k * self.headroom))


# --------------------------------------------------------------------------------------------------------------
def _reduce_scatter(out: torch.Tensor, flat: torch.Tensor, group):
    """out <- this rank's shard of sum_over_ranks(flat).  RCCL: one reduce_scatter; gloo (CPU rehearsals) has no
    reduce_scatter: all-reduce a copy and slice."""
    if dist.get_backend(group) == "gloo":
        tmp = flat.clone()
        dist.all_reduce(tmp, group=group)
        r, s = dist.get_rank(group), out.numel()
        out.copy_(tmp[r * s:(r + 1) * s])
        return None
    return dist.reduce_scatter_tensor(out, flat, group=group, async_op=True)


def _all_gather(flat: torch.Tensor, shard: torch.Tensor, group):
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(shard) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, shard.clone(), group=group)
        flat.copy_(torch.cat(parts))
        return
    dist.all_gather_into_tensor(flat, shard, group=group)       # in place: `shard` is this rank's slice of `flat`


class ShardedGrids:
    """The dense-grid parameters of a renderer as ONE flat buffer cut into world-size shards (option 2 of SURVEY
    8(e)).  ``names``: attribute names of DenseGrid modules, in the order of the step's flat gradient buffer
    (trainer.FineStep / LtsStep: sdf, off_color, emo_color[, brdf]).  After construction ``module.grid`` of each is a
    VIEW into ``self.flat`` with its original logical shape and storage order."""

    def __init__(self, model, names: List[str], process_group):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        params = [getattr(model, n).grid for n in names]
        self.names = names
        self.sizes = [p.numel() for p in params]
        self.shapes = [tuple(p.shape) for p in params]       # logical [1,C,X,Y,Z]; colour grids are STORED [1,X,Y,Z,C]
        n = sum(self.sizes)
        quantum = self.world * 128                           # whole 512-byte bricks per shard
        self.n = n
        self.padded = (n + quantum - 1) // quantum * quantum
        self.shard = self.padded // self.world
        dev = params[0].device
        self.flat = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        o = 0
        self.bounds: List[Tuple[str, int, int]] = []         # (name, begin, end) in the flat buffer
        with torch.no_grad():
            for name, p in zip(names, params):
                k = p.numel()
                seg = self.flat[o:o + k]
                if p.dim() == 5 and p.shape[1] > 1:          # channels-last storage: memory order [1,X,Y,Z,C]
                    seg.view(p.shape[0], *p.shape[2:], p.shape[1]).copy_(p.detach().permute(0, 2, 3, 4, 1))
                    p.data = seg.view(p.shape[0], *p.shape[2:], p.shape[1]).permute(0, 4, 1, 2, 3)
                else:
                    seg.view(p.shape).copy_(p.detach())
                    p.data = seg.view(p.shape)
                self.bounds.append((name, o, o + k))
                o += k
        self.grad_shard = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self._work = None

    def my_range(self) -> Tuple[int, int]:
        return self.rank * self.shard, (self.rank + 1) * self.shard

    # flat-buffer order <-> the reference's per-parameter layout (contiguous [1,C,X,Y,Z]): checkpoints interchange
    def to_reference_layout(self, flat: torch.Tensor):
        """[padded] buffer in this object's order -> {name: contiguous tensor of the parameter's logical shape}."""
        out = {}
        for (name, b, e), shp in zip(self.bounds, self.shapes):
            seg = flat[b:e]
            if len(shp) == 5 and shp[1] > 1:
                out[name] = seg.view(shp[0], *shp[2:], shp[1]).permute(0, 4, 1, 2, 3).contiguous()
            else:
                out[name] = seg.view(shp).clone()
        return out

    def from_reference_layout(self, tensors, device=None) -> torch.Tensor:
        """{name: tensor of the parameter's logical shape, any strides} -> [padded] buffer in this object's order."""
        flat = torch.zeros(self.padded, dtype=torch.float32, device=device or self.flat.device)
        for (name, b, e), shp in zip(self.bounds, self.shapes):
            t = tensors[name].to(flat.device, torch.float32)
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{name}: shape {tuple(t.shape)} does not match the parameter's {tuple(shp)}")
            if len(shp) == 5 and shp[1] > 1:
                flat[b:e].view(shp[0], *shp[2:], shp[1]).copy_(t.permute(0, 2, 3, 4, 1))
            else:
                flat[b:e].view(shp).copy_(t)
        return flat

    def reduce_scatter(self, flat_grad: torch.Tensor):
        """Start summing the grid part of the step's gradient buffer (``flat_grad``: [n] or [padded]) into
        ``grad_shard``; asynchronous on RCCL."""
        if flat_grad.numel() != self.padded:
            # (the trainer steps lay their flat buffer out with the grid part padded to ``self.padded`` once
            # ``step.sharded`` is attached -- trainer._grid_pad -- so this 218 MB-per-step copy is the fallback for other callers)
            if not hasattr(self, "_gpad") or self._gpad.device != flat_grad.device:
                self._gpad = torch.zeros(self.padded, dtype=torch.float32, device=flat_grad.device)
            self._gpad[: self.n].copy_(flat_grad[: self.n])
            flat_grad = self._gpad
        self._work = _reduce_scatter(self.grad_shard, flat_grad, self.pg)

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None

    def all_gather(self):
        lo, hi = self.my_range()
        _all_gather(self.flat, self.flat[lo:hi], self.pg)
