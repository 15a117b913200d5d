"""Spectral front end and loss reductions.

ROUND-1 STATUS: device ops expressed with PyTorch-ROCm (strided conv1d against the
windowed DFT basis, exactly the reference formulation, datasets/transforms.py:86-123);
the LDS radix-FFT kernel replaces ``stft_magnitude`` next (DESIGN.md "kernel status").
"""
import functools
import math

import torch
import torch.nn.functional as F


@functools.lru_cache(maxsize=16)
def _basis_cpu(n_fft, win_length):
    bins = n_fft // 2 + 1
    k = torch.arange(bins, dtype=torch.float64)[:, None]
    n = torch.arange(n_fft, dtype=torch.float64)[None, :]
    phase = 2.0 * math.pi * torch.remainder(k * n, n_fft) / n_fft
    basis = torch.cat([torch.cos(phase), -torch.sin(phase)], dim=0).to(torch.float32)
    m = torch.arange(win_length, dtype=torch.float64)
    hann = (0.5 - 0.5 * torch.cos(2.0 * math.pi * m / win_length)).to(torch.float32)
    window = torch.zeros(n_fft)
    lpad = (n_fft - win_length) // 2
    window[lpad:lpad + win_length] = hann
    return (basis * window)[:, None, :].contiguous()


_basis_dev = {}


def windowed_dft_basis(n_fft, win_length, device):
    key = (n_fft, win_length, str(device))
    if key not in _basis_dev:
        _basis_dev[key] = _basis_cpu(n_fft, win_length).to(device)
    return _basis_dev[key]


def stft_magnitude(x, n_fft, hop, win_length):
    """x [B, T] -> [B, n_fft/2+1, frames]."""
    b, t = x.shape
    pad = (n_fft - hop) // 2
    xp = F.pad(x.reshape(b, 1, 1, t), (pad, pad, 0, 0), mode="reflect").reshape(b, 1, t + 2 * pad)
    ft = F.conv1d(xp, windowed_dft_basis(n_fft, win_length, x.device), stride=hop)
    bins = n_fft // 2 + 1
    return torch.sqrt(ft[:, :bins] ** 2 + ft[:, bins:] ** 2)
