"""Test double of csrc/brick.hip in torch ops, used only by the CPU (gloo) rehearsals of the brick-sparse
gradient exchange and as the checker of the HIP kernels; never imported by the product path."""
import torch


class TorchBrickOps:
    brick = 128

    def _view(self, flat):
        n = flat.numel()
        nb = (n + self.brick - 1) // self.brick
        if n == nb * self.brick:
            return flat.view(nb, self.brick), None
        pad = torch.zeros(nb * self.brick, dtype=flat.dtype, device=flat.device)
        pad[:n] = flat
        return pad.view(nb, self.brick), n

    def flags(self, flat, out):
        v, _ = self._view(flat)
        out.copy_(v.ne(0).any(1).to(torch.uint8))

    def pack(self, flat, idx, packed):
        v, _ = self._view(flat)
        ok = idx >= 0                                    # negative = unused slot of a fixed-capacity list
        out = packed.view(-1, self.brick)
        out.zero_()
        out[ok] = v[idx[ok]]

    def unpack(self, packed, idx, flat):
        v, n = self._view(flat)
        ok = idx >= 0
        v[idx[ok]] = packed.view(-1, self.brick)[ok]
        if n is not None:
            flat.copy_(v.view(-1)[:n])
