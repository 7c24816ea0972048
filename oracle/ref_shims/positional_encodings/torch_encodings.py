"""Import shim restating positional_encodings==6.0.3 `PositionalEncoding3D` (third-party, pinned in the
reference's requirements.txt:3, call site busca/encodings.py:4,28-36).  The package is not installed in
the build container and there is no network, so this is a from-the-published-algorithm restatement:
per axis c = 2*ceil(d/6) channels of interleaved sin/cos of pos * 1/10000^(2j/c); the three axis blocks
are concatenated and truncated to d channels.  PARITY UNPINNED against the real package.
Only used to import the reference when generating golden vectors; never shipped to the GPU box path.
"""
import numpy as np
import torch
from torch import nn


def _interleaved(sin_inp):
    emb = torch.stack((sin_inp.sin(), sin_inp.cos()), dim=-1)
    return torch.flatten(emb, -2, -1)


class PositionalEncoding3D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.org_channels = channels
        channels = int(np.ceil(channels / 6) * 2)
        if channels % 2:
            channels += 1
        self.channels = channels
        inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
        self.register_buffer("inv_freq", inv_freq)
        self.register_buffer("cached_penc", None, persistent=False)

    def forward(self, tensor):
        if len(tensor.shape) != 5:
            raise RuntimeError("The input tensor has to be 5d!")
        _b, x, y, z, orig_ch = tensor.shape
        px = torch.arange(x, dtype=self.inv_freq.dtype)
        py = torch.arange(y, dtype=self.inv_freq.dtype)
        pz = torch.arange(z, dtype=self.inv_freq.dtype)
        ex = _interleaved(torch.einsum("i,j->ij", px, self.inv_freq)).unsqueeze(1).unsqueeze(1)
        ey = _interleaved(torch.einsum("i,j->ij", py, self.inv_freq)).unsqueeze(1)
        ez = _interleaved(torch.einsum("i,j->ij", pz, self.inv_freq))
        emb = torch.zeros((x, y, z, self.channels * 3), dtype=tensor.dtype)
        emb[:, :, :, : self.channels] = ex
        emb[:, :, :, self.channels: 2 * self.channels] = ey
        emb[:, :, :, 2 * self.channels:] = ez
        self.cached_penc = emb[None, :, :, :, :orig_ch]
        return self.cached_penc
