"""Spectral front end and spectral loss on the HIP library (include/smt_hip.h, "spectral"):
in-LDS FFT instead of the reference's DFT-as-conv1d (datasets/transforms.py:86-123).

Window and twiddle tables are computed once per (n_fft, win_length) in float64 and cached on
the device; the kernels never evaluate sin/cos.
"""
import math

import torch

from . import native as N
from . import profiler

_tables = {}


_consts = {}


def _const(values, device):
    """A small fp32 constant on the device, uploaded once per (values, device): a fresh torch.tensor(..., device=cuda) in
    every backward is a pageable host-to-device copy -- a launch more, and not capturable in a hipGraph."""
    key = (tuple(float(v) for v in values), str(device))
    if key not in _consts:
        _consts[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
    return _consts[key]


def _get_tables(n_fft, win_length, device):
    key = (n_fft, win_length, str(device))
    if key not in _tables:
        m = torch.arange(win_length, dtype=torch.float64)
        hann = 0.5 - 0.5 * torch.cos(2.0 * math.pi * m / win_length)       # periodic Hann (fftbins=True)
        window = torch.zeros(n_fft, dtype=torch.float64)
        lpad = (n_fft - win_length) // 2                                      # librosa.util.pad_center
        window[lpad:lpad + win_length] = hann
        k = torch.arange(n_fft // 2, dtype=torch.float64)
        ang = -2.0 * math.pi * k / n_fft
        tw = torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)
        _tables[key] = (window.to(torch.float32).to(device), tw.to(torch.float32).contiguous().to(device))
    return _tables[key]


def num_frames(t, n_fft, hop):
    return (t + 2 * ((n_fft - hop) // 2) - n_fft) // hop + 1


@torch.no_grad()
def stft_magnitude(x, n_fft, hop, win_length):
    """x [B, T] fp32 -> |STFT| [B, n_fft/2+1, frames] (STFT.forward, transforms.py:108-123)."""
    x = x.contiguous().float()
    b, t = x.shape
    window, tw = _get_tables(n_fft, win_length, x.device)
    frames = num_frames(t, n_fft, hop)
    mag = torch.empty(b, n_fft // 2 + 1, frames, dtype=torch.float32, device=x.device)
    with profiler.region("stft_mag", nbytes=x.numel() * 4 + mag.numel() * 4, bound="hbm"):
        N.check(N.lib().smt_stft_magnitude(N.ptr(x), N.ptr(window), N.ptr(tw), N.ptr(mag), b, t, n_fft, hop,
                                           N.stream_ptr()), "smt_stft_magnitude")
    return mag


@torch.no_grad()
def stft_inverse(magnitude, phase, n_fft, hop, win_length):
    """magnitude, phase [B, n_fft/2+1, frames] -> audio [B, 1, T'] (STFT.inverse, transforms.py:125-156)."""
    magnitude, phase = magnitude.contiguous().float(), phase.contiguous().float()
    b, bins, frames = magnitude.shape
    assert bins == n_fft // 2 + 1 and phase.shape == magnitude.shape
    window, tw = _get_tables(n_fft, win_length, magnitude.device)
    pad = (n_fft - hop) // 2
    out = torch.empty(b, (frames - 1) * hop + n_fft - 2 * pad, dtype=torch.float32, device=magnitude.device)
    with profiler.region("stft_inverse", nbytes=2 * magnitude.numel() * 4 + out.numel() * 4, bound="hbm"):
        N.check(N.lib().smt_stft_inverse(N.ptr(magnitude), N.ptr(phase), N.ptr(window), N.ptr(tw), N.ptr(out), b, n_fft, hop,
                                         frames, N.stream_ptr()), "smt_stft_inverse")
    return out.unsqueeze(1)


@torch.no_grad()
def log_mel(x, mel_basis, band, n_fft, hop, win_length):
    """x [B, T] -> log-mel [B, n_mels, frames] in one kernel (5.25 B/sample algorithmic)."""
    x = x.contiguous().float()
    b, t = x.shape
    window, tw = _get_tables(n_fft, win_length, x.device)
    frames = num_frames(t, n_fft, hop)
    n_mels = mel_basis.shape[0]
    mel = torch.empty(b, n_mels, frames, dtype=torch.float32, device=x.device)
    with profiler.region("melspec", nbytes=x.numel() * 4 + mel.numel() * 4, bound="hbm"):
        N.check(N.lib().smt_melspec(N.ptr(x), N.ptr(window), N.ptr(tw), N.ptr(mel_basis), N.ptr(band), N.ptr(mel), b, t,
                                    n_fft, hop, n_mels, N.stream_ptr()), "smt_melspec")
    return mel


class _StftLoss(torch.autograd.Function):
    """mean_b sqrt(sum ((|Y|-|Yh|) m)^2) [+ the same on clamped log magnitudes] for ONE resolution."""

    @staticmethod
    def forward(ctx, y, yh, lens, n_fft, hop, win_length, use_log):
        y, yh = y.contiguous().float(), yh.contiguous().float()
        b, t = y.shape
        window, tw = _get_tables(n_fft, win_length, y.device)
        frames = num_frames(t, n_fft, hop)
        part = torch.empty(b, frames, 2, dtype=torch.float32, device=y.device)
        lens32 = None if lens is None else lens.to(torch.int32)
        with profiler.region("stft_loss_fwd", nbytes=2 * y.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_stft_loss_fwd(N.ptr(y), N.ptr(yh), N.ptr(lens32), N.ptr(window), N.ptr(tw),
                                              N.ptr(part), b, t, n_fft, hop, N.stream_ptr()), "smt_stft_loss_fwd")
        sums = part.sum(dim=1)                       # [B, 2]; fixed-order reduction of tiny tensors
        root = sums.sqrt()
        loss = root[:, 0].mean() + (root[:, 1].mean() if use_log else 0.0)
        ctx.save_for_backward(y, yh, lens32 if lens32 is not None else torch.empty(0), root)
        ctx.cfg = (n_fft, hop, win_length, use_log, lens is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        y, yh, lens32, root = ctx.saved_tensors
        n_fft, hop, win_length, use_log, has_lens = ctx.cfg
        lens32 = lens32 if has_lens else None
        b, t = y.shape
        window, tw = _get_tables(n_fft, win_length, y.device)
        coef = (g / (2.0 * b)) / root.clamp(min=1e-30)          # d sqrt(S)/dS = 1/(2 sqrt(S)), mean over B
        if not use_log:
            coef = coef * _const([1.0, 0.0], coef.device)
        coef = coef.contiguous().float()
        dyh = torch.empty_like(yh)
        ws_bytes = N.lib().smt_stft_loss_bwd_workspace_bytes(b, t, n_fft, hop)
        ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=y.device)     # per-frame gradient rows
        with profiler.region("stft_loss_bwd", nbytes=3 * y.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_stft_loss_bwd(N.ptr(y), N.ptr(yh), N.ptr(lens32), N.ptr(window), N.ptr(tw),
                                              N.ptr(coef), N.ptr(dyh), b, t, n_fft, hop, N.ptr(ws), ws.numel(),
                                              N.stream_ptr()), "smt_stft_loss_bwd")
        return None, dyh, None, None, None, None, None


def stft_loss(y, yh, lens, n_fft, hop, win_length, use_log):
    return _StftLoss.apply(y, yh, lens, n_fft, hop, win_length, use_log)


class _ReconLoss(torch.autograd.Function):
    """l1 * mean|d| + l2 * mean d^2 + linf * sum_j mean_b topk_j(d^2) on masked [B, T] signals (losses.py:73-80)."""

    @staticmethod
    def forward(ctx, y, yh, lens, l1, l2, linf, topk):
        y, yh = y.contiguous().float(), yh.contiguous().float()
        b, t = y.shape
        lens32 = None if lens is None else lens.to(torch.int32)
        stats = torch.empty(b, 8, dtype=torch.float32, device=y.device)
        with profiler.region("recon_loss_fwd", nbytes=5 * 2 * y.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_recon_loss_fwd(N.ptr(y), N.ptr(yh), N.ptr(lens32), b, t, int(topk), N.ptr(stats),
                                               N.stream_ptr()), "smt_recon_loss_fwd")
        sums = stats[:, :3].sum(dim=0)                 # fixed-order reduction over the batch
        loss = l2 * sums[0] / (b * t) + linf * sums[2] / b
        if l1:
            loss = loss + l1 * sums[1] / (b * t)
        ctx.save_for_backward(y, yh, lens32 if lens32 is not None else torch.empty(0), stats)
        ctx.cfg = (l1, l2, linf, int(topk), lens is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        y, yh, lens32, stats = ctx.saved_tensors
        l1, l2, linf, topk, has_lens = ctx.cfg
        lens32 = lens32 if has_lens else None
        b, t = y.shape
        scale = _const([l1 / (b * t), 2.0 * l2 / (b * t), 2.0 * linf / b], y.device)
        coef = (g.reshape(1).float() * scale).contiguous()
        dyh = torch.empty_like(yh)
        with profiler.region("recon_loss_bwd", nbytes=3 * y.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_recon_loss_bwd(N.ptr(y), N.ptr(yh), N.ptr(lens32), N.ptr(stats), N.ptr(coef), b, t, topk,
                                               N.ptr(dyh), N.stream_ptr()), "smt_recon_loss_bwd")
        return None, dyh, None, None, None, None, None


def recon_loss(y, yh, lens, l1, l2, linf, topk):
    return _ReconLoss.apply(y, yh, lens, float(l1), float(l2), float(linf), int(topk))
