"""Helpers of the GlowTTS package that have native counterparts (reference models/glow_tts/submodules.py):
``sequence_mask`` (:18-25, used at models/vqvae/vqvae.py:99) and the monotonic alignment search ``maximum_path`` (:28-67)."""
import math

import torch

from smt_amd import native as N
from smt_amd import profiler


def sequence_mask(length, max_length=None):
    if max_length is None:
        max_length = int(length.max())
    steps = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return steps[None, :] < length[:, None]


@torch.no_grad()
def maximum_path(value, mask, max_neg_val=None):
    """Monotonic alignment search (submodules.py:28-67): value, mask [b, t_x, t_y] -> 0/1 path of the same shape, device and
    dtype.  One HIP kernel (smt_maximum_path) instead of the reference's device -> host -> numpy -> device round trip."""
    if max_neg_val is None:
        max_neg_val = -math.inf
    assert value.is_cuda and value.dim() == 3 and mask.shape == value.shape
    b, t_x, t_y = value.shape
    v32, m32 = value.detach().float().contiguous(), mask.detach().float().contiguous()
    path = torch.empty_like(v32)
    with profiler.region("maximum_path", nbytes=3 * v32.numel() * 4, bound="hbm"):
        N.check(N.lib().smt_maximum_path(N.ptr(v32), N.ptr(m32), b, t_x, t_y, float(max_neg_val), N.ptr(path),
                                         N.stream_ptr()), "smt_maximum_path")
    return path.to(value.dtype)
