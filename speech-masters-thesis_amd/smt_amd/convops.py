"""Channels-last ("NTC", [B, T, C]) building blocks of the encoder / decoder stacks.

This module is the single seam between the model code (models/vqvae/*) and the
arithmetic.  ROUND-1 STATUS: the VQ, EMA and loss reductions run in libsmt_hip.so;
the convolution entry points below are still expressed with PyTorch-ROCm device ops
(MIOpen) and are being replaced one by one by the hand-written MFMA kernels -- see
DESIGN.md "kernel status".  They run on the GPU only; nothing here touches the
oracle or a CPU path.
"""
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


@dataclass
class DropSpec:
    """relu(dropout(x)) prologue of a conv (reference models/vqvae/resnet.py:22-26)."""
    p: float
    training: bool
    seed: int = 0
    site: int = 0


def row_mask(lens: torch.Tensor, t: int, dtype) -> torch.Tensor:
    """[B, T, 1] prefix mask: 1 where t < lens[b]."""
    steps = torch.arange(t, device=lens.device)
    return (steps[None, :] < lens[:, None]).to(dtype).unsqueeze(-1)


def _prologue(x, lens, act: Optional[DropSpec]):
    if lens is not None:
        x = x * row_mask(lens, x.shape[1], x.dtype)
    if act is not None:
        x = torch.relu(F.dropout(x, p=act.p, training=act.training))
    return x


def conv1d(x, weight, bias, *, stride=1, padding=0, dilation=1, lens=None, act=None, residual=None):
    """y[b,t,:] = sum_j W_j . pro(x)[b, t*stride + j*dilation - padding, :] + bias (+ residual).
    ``weight`` keeps torch's Conv1d layout [Cout, Cin, k] (checkpoint compatibility)."""
    x = _prologue(x, lens, act)
    y = F.conv1d(x.transpose(1, 2), weight.to(x.dtype), bias.to(x.dtype), stride=stride, padding=padding,
                 dilation=dilation).transpose(1, 2)
    if residual is not None:
        y = y + residual
    return y


def conv_transpose1d(x, weight, bias, *, stride, padding, lens=None):
    """ConvTranspose1d with torch's [Cin, Cout, k] weight layout."""
    x = _prologue(x, lens, None)
    return F.conv_transpose1d(x.transpose(1, 2), weight.to(x.dtype), bias.to(x.dtype), stride=stride,
                              padding=padding).transpose(1, 2)


def gate_mix(z, depth: int):
    """sum_d tanh(t_d) * softmax_d(s_d) over the ``depth`` branches laid side by side in the channel
    dimension: z = [.., d*(2w) + (0..w-1)] = t_d, [.., d*(2w) + (w..2w-1)] = s_d
    (reference models/vqvae/resnet.py:229-237)."""
    b, t, c = z.shape
    w = c // (2 * depth)
    z = z.view(b, t, depth, 2, w)
    return (torch.tanh(z[:, :, :, 0]) * torch.softmax(z[:, :, :, 1].float(), dim=2).to(z.dtype)).sum(dim=2)
