"""CPU restatement (torch-CPU / numpy, fp32 unless stated) of the reference
VQ-VAE hot path.  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Layout conventions are the reference's: activations ``[B, C, T]`` (NCT),
parameters in a flat ``dict`` keyed by the reference ``state_dict`` names
(``encoders.0.level_blocks.0.blocks.0.weight`` ...), so a captured reference
state dict drives this code unchanged.

Citations are ``file:line`` relative to the reference repository root.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

SAFE_LOG_EPS = 1e-5  # utils/torch_utils.py:4


def safe_log(x: Tensor, eps: float = SAFE_LOG_EPS) -> Tensor:
    """utils/torch_utils.py:4-5 -- log(clamp(x, min=eps))."""
    return torch.log(torch.clamp(x, min=eps))


def sequence_mask(length: Tensor, max_length: Optional[int] = None) -> Tensor:
    """models/glow_tts/submodules.py:18-25 -- arange(max) < len[:, None] (bool)."""
    if max_length is None:
        max_length = int(length.max())
    pos = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return pos[None, :] < length[:, None]


def window_sumsquare(n_fft: int, win_length: int, hop: int, n_frames: int) -> np.ndarray:
    """librosa.filters.window_sumsquare(window="hann", n_frames, hop_length, win_length, n_fft, norm=None), restated from
    its documented algorithm (librosa is absent offline -- parity unpinned for this helper): the squared periodic Hann
    window, centre-padded to n_fft, summed at every frame offset; length n_fft + hop * (n_frames - 1)."""
    win_sq = stft_window(n_fft, win_length).astype(np.float32) ** 2
    n = n_fft + hop * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    for i in range(n_frames):
        sample = i * hop
        x[sample:min(n, sample + n_fft)] += win_sq[:max(0, min(n_fft, n - sample))]
    return x


def stft_inverse(magnitude: Tensor, phase: Tensor, n_fft: int, hop: int, win_length: int, sumsquare=None) -> Tensor:
    """STFT.inverse (transforms.py:125-156) with the inverse basis of STFT.__init__ (:86-105), same operations in the same
    order: recombine, conv_transpose1d with the windowed pseudo-inverse basis, divide by the window sum-square where it
    exceeds float tiny, times n_fft / hop, trim pad_amount on both sides."""
    scale = n_fft / hop
    pad = (n_fft - hop) // 2
    fourier = np.fft.fft(np.eye(n_fft))
    cutoff = n_fft // 2 + 1
    basis = np.vstack([np.real(fourier[:cutoff]), np.imag(fourier[:cutoff])])
    inv = torch.tensor(np.linalg.pinv(scale * basis).T[:, None, :], dtype=torch.float32)
    inv = inv * torch.from_numpy(np.asarray(stft_window(n_fft, win_length))).float()
    rec = torch.cat([magnitude * torch.cos(phase), magnitude * torch.sin(phase)], dim=1)
    out = F.conv_transpose1d(rec, inv, stride=hop, padding=0)
    wss = (sumsquare or window_sumsquare)(n_fft, win_length, hop, magnitude.size(-1))
    nz = torch.from_numpy(np.where(wss > np.finfo(np.float32).tiny)[0])
    out[:, :, nz] /= torch.from_numpy(wss)[nz]
    out *= scale
    out = out[:, :, pad:]
    return out[:, :, :-pad]


# ---------------------------------------------------------------------------
# STFT / mel front end  (datasets/transforms.py)
# ---------------------------------------------------------------------------

def hann_periodic(win_length: int) -> np.ndarray:
    """scipy.signal.get_window('hann', M, fftbins=True) restated
    (datasets/transforms.py:97): 0.5 - 0.5 cos(2 pi n / M), n = 0..M-1."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def stft_window(n_fft: int, win_length: int) -> np.ndarray:
    """Hann window centre-padded with zeros to n_fft (transforms.py:97-99;
    librosa.util.pad_center: lpad = (n_fft - win) // 2).  float32."""
    w = np.zeros(n_fft, dtype=np.float64)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w.astype(np.float32)


def stft_forward_basis(n_fft: int, win_length: int) -> Tensor:
    """Windowed real-DFT basis ``[n_fft + 2, 1, n_fft]`` (transforms.py:88-105):
    rows 0..n_fft/2 are cos(2 pi k n / N), rows n_fft/2+1.. are -sin(...),
    cast to fp32 and THEN multiplied by the fp32 window (order matters for the
    last bit)."""
    cutoff = n_fft // 2 + 1
    k = np.arange(cutoff, dtype=np.float64)[:, None]
    n = np.arange(n_fft, dtype=np.float64)[None, :]
    # exact integer phase reduction keeps cos/sin accurate for large k*n
    phase = 2.0 * np.pi * ((k * n) % n_fft) / n_fft
    basis = np.vstack([np.cos(phase), -np.sin(phase)]).astype(np.float32)
    basis = torch.from_numpy(basis)[:, None, :]
    return basis * torch.from_numpy(stft_window(n_fft, win_length))


def stft_num_frames(num_samples: int, n_fft: int, hop: int) -> int:
    pad = (n_fft - hop) // 2
    return (num_samples + 2 * pad - n_fft) // hop + 1


def stft_magnitude(x: Tensor, n_fft: int, hop: int, win_length: int,
                   basis: Optional[Tensor] = None) -> Tensor:
    """STFT.forward (transforms.py:108-123): reflect-pad (n_fft-hop)//2 each
    side, strided conv1d with the windowed DFT basis, sqrt(re^2 + im^2).
    ``x`` is ``[B, T]`` or ``[B, 1, T]``; returns ``[B, n_fft/2+1, frames]``."""
    if basis is None:
        basis = stft_forward_basis(n_fft, win_length)
    b, t = x.shape[0], x.shape[-1]
    pad = (n_fft - hop) // 2
    xp = F.pad(x.reshape(b, 1, 1, t), (pad, pad, 0, 0), mode="reflect").reshape(b, 1, t + 2 * pad)
    ft = F.conv1d(xp, basis.to(x.dtype), stride=hop)
    cutoff = n_fft // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    return torch.sqrt(re * re + im * im)


def _hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sample_rate: int = 22050, n_fft: int = 1024, n_mels: int = 80,
                   f_min: float = 0.0, f_max: Optional[float] = 8000.0) -> np.ndarray:
    """Restatement of librosa.filters.mel's documented defaults (Slaney mel
    scale, slaney area normalisation) -- call site transforms.py:38-44, params
    ljspeech.py:58-66.  librosa is NOT in the reference tree and its version is
    unpinned (requirements.txt:5): **parity unpinned** for this function; the
    mel golden is pinned to this restatement."""
    if f_max is None:
        f_max = sample_rate / 2.0
    n_bins = n_fft // 2 + 1
    fft_freqs = np.linspace(0.0, sample_rate / 2.0, n_bins)
    mel_pts = np.linspace(_hz_to_mel_slaney(f_min), _hz_to_mel_slaney(f_max), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def mel_spectrogram(audio: Tensor, mel_basis: Tensor, n_fft: int = 1024, hop: int = 256,
                    win_length: int = 1024) -> Tensor:
    """MelSpectrogram.forward (transforms.py:48-65) without the unused jitter
    branch: range assert, |STFT| (magnitude, not power), mel matmul, safe_log."""
    assert audio.min() >= -1 and audio.max() <= 1  # transforms.py:49
    if audio.dim() == 1:
        audio = audio[None]
    mag = stft_magnitude(audio, n_fft, hop, win_length)
    return safe_log(torch.matmul(mel_basis, mag))


# ---------------------------------------------------------------------------
# Model configuration (configs/models/vqvae.yaml, after the vqvae.py:65-70 HACK)
# ---------------------------------------------------------------------------

@dataclass
class VQVAEConfig:
    levels: int = 3
    downs_t: Tuple[int, ...] = (3, 2, 2)
    strides_t: Tuple[int, ...] = (2, 2, 2)
    emb_width: int = 128
    l_bins: int = 512
    mu: float = 0.99
    multipliers: Tuple[int, ...] = (2, 1, 1)
    width: int = 64
    depth: int = 4
    revival_threshold: float = 1.0
    dilation_growth_rate: int = 3
    dilation_cycle: Optional[int] = None
    kernel_size_growth_rate: int = 2
    kernel_size_cycle: Optional[int] = None
    zero_out: bool = True
    dropout: float = 0.1            # resnet.py:18 default, not in the yaml
    commit: float = 0.05
    multispectral: float = 1.0
    l1: float = 0.0
    l2: float = 1.0
    linf: float = 0.02
    linf_topk: int = 2048
    n_ffts: Tuple[int, ...] = (2048, 1024, 512)
    hop_lengths: Tuple[int, ...] = (240, 120, 50)
    win_lengths: Tuple[int, ...] = (1200, 600, 240)
    log_stft: bool = True

    # effective single level kept by vqvae.py:65-70 (LEVEL = -1)
    @property
    def eff_width(self) -> int:
        return self.width * self.multipliers[-1]

    @property
    def eff_depth(self) -> int:
        return self.depth * self.multipliers[-1]

    @property
    def compression(self) -> int:
        c = 1
        for d, s in zip(self.downs_t, self.strides_t):
            c *= s ** d
        return c

    @staticmethod
    def from_dict(model_cfg: dict) -> "VQVAEConfig":
        loss = model_cfg.get("loss", {})
        return VQVAEConfig(
            levels=int(model_cfg["levels"]),
            downs_t=tuple(model_cfg["downs_t"]),
            strides_t=tuple(model_cfg["strides_t"]),
            emb_width=int(model_cfg["emb_width"]),
            l_bins=int(model_cfg["l_bins"]),
            mu=float(model_cfg["mu"]),
            multipliers=tuple(model_cfg["multipliers"] or [1] * int(model_cfg["levels"])),
            width=int(model_cfg["width"]),
            depth=int(model_cfg["depth"]),
            revival_threshold=float(model_cfg["revival_threshold"]),
            dilation_growth_rate=int(model_cfg["dilation_growth_rate"]),
            dilation_cycle=model_cfg.get("dilation_cycle"),
            kernel_size_growth_rate=int(model_cfg["kernel_size_growth_rate"]),
            kernel_size_cycle=model_cfg.get("kernel_size_cycle"),
            zero_out=bool(model_cfg["zero_out"]),
            commit=float(loss["commit"]), multispectral=float(loss["multispectral"]),
            l1=float(loss["l1"]), l2=float(loss["l2"]), linf=float(loss["linf"]),
            linf_topk=int(loss["linf_topk"]),
            n_ffts=tuple(loss["n_ffts"]), hop_lengths=tuple(loss["hop_lengths"]),
            win_lengths=tuple(loss["win_lengths"]), log_stft=bool(loss["log"]),
        )


def _mod_cycle(depth: int, cycle: Optional[int]) -> int:
    """models/vqvae/resnet.py:9-13."""
    return depth if cycle is None else depth % cycle


def branch_geometry(cfg: VQVAEConfig, d: int) -> Tuple[int, int, int]:
    """(kernel, dilation, padding) of GatedHiFi branch d -- resnet.py:209-210, :20."""
    dil = cfg.dilation_growth_rate ** _mod_cycle(d, cfg.dilation_cycle)
    k = 3 + cfg.kernel_size_growth_rate * _mod_cycle(d, cfg.kernel_size_cycle)
    return k, dil, ((k - 1) * dil) // 2


# Dropout is RNG-dependent in the reference (torch global RNG, resnet.py:22,25).
# The oracle takes the keep-mask as an explicit function so that the product's
# counter-based generator can be restated bit-exactly (see ``dropout_mask_ntc``).
DropFn = Callable[[str, Tensor], Tensor]


def no_dropout(site: str, x: Tensor) -> Tensor:
    return x


# ---------------------------------------------------------------------------
# Conv stacks (models/vqvae/conv.py, resnet.py, encdec.py)
# ---------------------------------------------------------------------------

def masked_conv1d(x, mask, w, b, stride=1, padding=0, dilation=1):
    """MaskedConv1d.forward (conv.py:7-10): conv(x*mask); mask[:, :, ::stride]."""
    y = F.conv1d(x * mask, w, b, stride=stride, padding=padding, dilation=dilation)
    return y, mask[:, :, ::stride]


def masked_conv_transpose1d(x, mask, w, b, stride, padding):
    """MaskedConvTranspose1d.forward (conv.py:15-18)."""
    y = F.conv_transpose1d(x * mask, w, b, stride=stride, padding=padding)
    return y, mask.repeat_interleave(stride, dim=-1)


def res_layer(x, p: Params, prefix: str, k: int, dil: int, pad: int, res_scale: float,
              drop: DropFn) -> Tensor:
    """ResLayer.forward (resnet.py:16-36):
    x + res_scale * Conv1x1(ReLU(Drop(Conv_k,dil(ReLU(Drop(x))))))."""
    h = torch.relu(drop(prefix + ".drop0", x))
    h = F.conv1d(h, p[prefix + ".model.2.weight"], p[prefix + ".model.2.bias"], padding=pad, dilation=dil)
    h = torch.relu(drop(prefix + ".drop1", h))
    h = F.conv1d(h, p[prefix + ".model.5.weight"], p[prefix + ".model.5.bias"])
    return x + res_scale * h


def gated_hifi_block(x, mask, p: Params, prefix: str, cfg: VQVAEConfig, drop: DropFn) -> Tensor:
    """GatedHiFiBlock.forward (resnet.py:222-241).  res_scale=False => 1.0
    (conv.py:55, resnet.py:201)."""
    ts, ss = [], []
    xm = x * mask
    for d in range(cfg.eff_depth):
        k, dil, pad = branch_geometry(cfg, d)
        bp = f"{prefix}.blocks.{d}"
        h = F.conv1d(xm, p[bp + ".0.weight"], p[bp + ".0.bias"])
        z = res_layer(h, p, bp + ".1", k, dil, pad, 1.0, drop)
        z_t, z_s = z.chunk(2, dim=1)
        ts.append(z_t)
        ss.append(z_s)
    t = torch.stack(ts, dim=1)
    s = torch.stack(ss, dim=1)
    z = (torch.tanh(t) * torch.softmax(s, dim=1)).sum(dim=1)
    z = F.conv1d(z * mask, p[prefix + ".gate.weight"], p[prefix + ".gate.bias"])
    return x + z


def encoder_forward(x, mask, p: Params, cfg: VQVAEConfig, drop: DropFn = no_dropout,
                    prefix: str = "encoders.0"):
    """Encoder.forward / EncoderConvBlock (encdec.py:28-40, conv.py:38-84): per
    level ``down_t`` x [strided conv k=2s, s, p=s//2 -> GatedHiFi] then conv k3."""
    for level in range(cfg.levels):
        down_t, stride_t = cfg.downs_t[level], cfg.strides_t[level]
        lp = f"{prefix}.level_blocks.{level}.blocks"
        i = 0
        for _ in range(down_t):
            x, mask = masked_conv1d(x, mask, p[f"{lp}.{i}.weight"], p[f"{lp}.{i}.bias"],
                                    stride=stride_t, padding=stride_t // 2)
            x = gated_hifi_block(x, mask, p, f"{lp}.{i + 1}", cfg, drop)
            i += 2
        x, mask = masked_conv1d(x, mask, p[f"{lp}.{i}.weight"], p[f"{lp}.{i}.bias"], padding=1)
    return x, mask


def decoder_forward(x, mask, p: Params, cfg: VQVAEConfig, drop: DropFn = no_dropout,
                    prefix: str = "decoders.0"):
    """Decoder.forward with all_levels=False / DecoderConvBlock (encdec.py:63-83,
    conv.py:87-143): levels reversed; conv k3 -> down_t x [GatedHiFi -> ConvT];
    final 1x1 conv on x*mask (encdec.py:82)."""
    for level in reversed(range(cfg.levels)):
        down_t, stride_t = cfg.downs_t[level], cfg.strides_t[level]
        lp = f"{prefix}.level_blocks.{level}.blocks"
        x, mask = masked_conv1d(x, mask, p[f"{lp}.0.weight"], p[f"{lp}.0.bias"], padding=1)
        i = 1
        for _ in range(down_t):
            x = gated_hifi_block(x, mask, p, f"{lp}.{i}", cfg, drop)
            x, mask = masked_conv_transpose1d(x, mask, p[f"{lp}.{i + 1}.weight"], p[f"{lp}.{i + 1}.bias"],
                                              stride=stride_t, padding=stride_t // 2)
            i += 2
    y = F.conv1d(x * mask, p[prefix + ".out.weight"], p[prefix + ".out.bias"])
    return y, mask


# ---------------------------------------------------------------------------
# Vector quantiser (models/vqvae/bottleneck.py)
# ---------------------------------------------------------------------------

def vq_preprocess(x: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """BottleneckBlock.preprocess (bottleneck.py:92-116) minus the unused
    ``prenorm``: NCT -> [N*T, C]; mask -> [N*T, 1]."""
    xf = x.permute(0, 2, 1).contiguous().view(-1, x.shape[1])
    mf = mask.permute(0, 2, 1).contiguous().reshape(-1, 1)
    return xf, mf


def vq_distance_fp32(x: Tensor, k: Tensor) -> Tensor:
    """bottleneck.py:128-133 verbatim arithmetic: sum(x^2) - 2 x k^T + sum(k^2)."""
    k_w = k.t()
    return (x ** 2).sum(dim=-1, keepdim=True) - 2 * torch.matmul(x, k_w) + (k_w ** 2).sum(dim=0, keepdim=True)


def vq_quantize_reference(x: Tensor, k: Tensor, mask: Optional[Tensor] = None):
    """BottleneckBlock.quantize (bottleneck.py:126-141) in the reference's own
    fp32 arithmetic, INCLUDING the [N]*[N,1] -> [N,N] broadcast in the masked
    fit (= sum over ALL rows of min_distance / K).  Returns (idx, fit, min_d)."""
    dist = vq_distance_fp32(x, k)
    min_d, idx = torch.min(dist, dim=-1)
    if mask is None:
        fit = min_d.mean()
    else:
        # (min_d[N] * mask[N,1]).sum() == min_d.sum() * mask.sum(); the mask sums cancel
        fit = (min_d.sum() * mask.sum()) / (mask.sum() * dist.shape[-1])
    return idx, fit, min_d


def vq_argmin_exact(x: np.ndarray, k: np.ndarray, chunk: int = 4096):
    """THE index semantics of this build: the exact argmin of ||x - k_j||^2 over
    the fp32 inputs, evaluated in float64 as sum_i (x_i - k_ji)^2 accumulated
    in index order, first (lowest) index on ties (torch.min tie rule,
    bottleneck.py:134).  The reference's fp32 sgemm expression agrees with this
    on every row whose best/runner-up margin is not at fp32 round-off level
    ('Bit-exact argmin'); goldens record the margins.

    Returns (idx int64 [N], d_best float64 [N], d_second float64 [N])."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    n = x.shape[0]
    idx = np.empty(n, dtype=np.int64)
    d1 = np.empty(n, dtype=np.float64)
    d2 = np.empty(n, dtype=np.float64)
    for s in range(0, n, chunk):
        xs = x[s:s + chunk]
        acc = np.zeros((xs.shape[0], k.shape[0]), dtype=np.float64)
        for i in range(x.shape[1]):  # index-order accumulation
            diff = xs[:, i:i + 1] - k[None, :, i]
            acc += diff * diff
        best = np.argmin(acc, axis=1)  # first minimum
        rows = np.arange(xs.shape[0])
        idx[s:s + chunk] = best
        d1[s:s + chunk] = acc[rows, best]
        if k.shape[0] > 1:
            acc[rows, best] = np.inf
            d2[s:s + chunk] = acc.min(axis=1)
        else:
            d2[s:s + chunk] = np.inf
    return idx, d1, d2


def vq_tile(x: Tensor, k_bins: int, noise: Optional[Tensor] = None) -> Tensor:
    """BottleneckBlock._tile (bottleneck.py:26-33).  ``noise`` (standard normal,
    shape of the repeated tensor) replaces the reference's randn_like."""
    d, ew = x.shape
    if d < k_bins:
        n_rep = (k_bins + d - 1) // d
        std = 0.01 / np.sqrt(ew)
        x = x.repeat(n_rep, 1)
        if noise is None:
            noise = torch.randn_like(x)
        x = x + noise * std
    return x


@dataclass
class CodebookState:
    """bottleneck.py:20-24,45-46: buffer ``k`` plus the plain attributes."""
    k: Tensor
    k_sum: Optional[Tensor] = None
    k_elem: Optional[Tensor] = None
    init: bool = False


def vq_init_k(state: CodebookState, k_rand: Tensor) -> None:
    """init_k (bottleneck.py:35-46) with the random rows supplied."""
    state.init = True
    state.k = k_rand.clone()
    state.k_sum = state.k.clone()
    state.k_elem = torch.ones(state.k.shape[0], dtype=state.k.dtype)


def vq_update_k(state: CodebookState, x: Tensor, x_l: Tensor, k_rand: Tensor, mu: float,
                threshold: float, reduce_fn: Optional[Callable[[Tensor, Tensor], None]] = None):
    """update_k (bottleneck.py:60-90).  ``x`` / ``x_l`` are the UNMASKED rows
    and their codes; ``k_rand`` replaces ``y[randperm][:K]`` (rank 0's after the
    broadcast at :73).  ``reduce_fn(_k_sum, _k_elem)`` stands for the two
    all-reduces at :74-75.  Returns the metrics dict of :85-90."""
    k_bins, emb = state.k.shape
    with torch.no_grad():
        onehot = torch.zeros(k_bins, x.shape[0], dtype=x.dtype)
        onehot.scatter_(0, x_l.view(1, -1), 1)
        _k_sum = onehot @ x
        _k_elem = onehot.sum(dim=-1)
        if reduce_fn is not None:
            reduce_fn(_k_sum, _k_elem)
        old_k = state.k
        state.k_sum = mu * state.k_sum + (1.0 - mu) * _k_sum
        state.k_elem = mu * state.k_elem + (1.0 - mu) * _k_elem
        usage = (state.k_elem.view(k_bins, 1) >= threshold).float()
        state.k = usage * (state.k_sum.view(k_bins, emb) / state.k_elem.view(k_bins, 1)) + (1 - usage) * k_rand
        _k_prob = _k_elem / _k_elem.sum()
        entropy = -(_k_prob * safe_log(_k_prob)).sum()
        used_curr = (_k_elem >= threshold).sum()
        usage_n = usage.sum()
        dk = torch.norm(state.k - old_k) / np.sqrt(np.prod(old_k.shape))
    return dict(entropy=entropy, used_curr=used_curr, usage=usage_n, dk=dk), _k_sum, _k_elem


def vq_forward(x: Tensor, mask: Tensor, state: CodebookState, mu: float, threshold: float,
               update_k: bool, k_rand: Optional[Tensor] = None, exact_indices: bool = True,
               reduce_fn=None):
    """BottleneckBlock.forward (bottleneck.py:171-201).

    ``exact_indices=True`` uses ``vq_argmin_exact`` (this build's semantics);
    ``False`` uses the reference's fp32 expression.  ``k_rand`` supplies the
    random rows for init_k / update_k."""
    n, _, t = x.shape
    xf, mf = vq_preprocess(x, mask)
    sel = (mf != 0)[:, 0]
    if callable(k_rand):
        # provider form: rows -> [K, D] (stands for y[randperm][:K], bottleneck.py:40,70)
        k_rand_fn = k_rand
    else:
        k_rand_fn = (lambda rows: k_rand)
    if update_k and not state.init:
        assert k_rand is not None
        vq_init_k(state, k_rand_fn(xf[sel].detach()))
    with torch.no_grad():
        ref_idx, fit, _ = vq_quantize_reference(xf, state.k, mf)
        if exact_indices:
            idx_np, _, _ = vq_argmin_exact(xf.detach().numpy(), state.k.numpy())
            x_l = torch.from_numpy(idx_np)
        else:
            x_l = ref_idx
        x_d = F.embedding(x_l, state.k)
    metrics = {}
    if update_k:
        assert k_rand is not None
        metrics, _, _ = vq_update_k(state, xf[sel].detach(), x_l[sel], k_rand_fn(xf[sel].detach()), mu,
                                    threshold, reduce_fn)
    commit = torch.norm(x_d[sel].detach() - xf[sel]) ** 2 / (mf.sum() * xf.shape[1])
    x_st = xf + (x_d - xf).detach()
    x_dq = x_st.view(n, t, -1).permute(0, 2, 1).contiguous()
    return x_l.view(n, t), x_dq * mask, commit, dict(fit=fit, **metrics)


# ---------------------------------------------------------------------------
# Losses (models/vqvae/losses.py)
# ---------------------------------------------------------------------------

def downsample_mask(mask: Tensor, n_fft: int, hop: int) -> Tensor:
    """MultiResolutionSpectralLoss.downsample_mask (losses.py:33-37)."""
    pad = (n_fft - hop) // 2
    m = F.pad(mask, (pad, 0), value=1.0)
    m = F.pad(m, (0, pad), value=0.0)
    return m[:, :, n_fft // 2: -n_fft // 2 + 1: hop]


def multires_stft_loss(y: Tensor, yh: Tensor, mask: Tensor, cfg: VQVAEConfig) -> Tensor:
    """MultiResolutionSpectralLoss.forward (losses.py:39-55)."""
    loss = 0.0
    for n_fft, hop, win in zip(cfg.n_ffts, cfg.hop_lengths, cfg.win_lengths):
        basis = stft_forward_basis(n_fft, win)
        ys = stft_magnitude(y, n_fft, hop, win, basis)
        yhs = stft_magnitude(yh, n_fft, hop, win, basis)
        m = downsample_mask(mask, n_fft, hop)
        loss = loss + ((ys * m - yhs * m) ** 2).sum(-1).sum(-1).sqrt().mean(0)
        if cfg.log_stft:
            loss = loss + ((safe_log(ys) * m - safe_log(yhs) * m) ** 2).sum(-1).sum(-1).sqrt().mean(0)
    return loss / len(cfg.n_ffts)


def multinorm_recon_loss(y: Tensor, yh: Tensor, mask: Tensor, cfg: VQVAEConfig) -> Tensor:
    """MultiNormReconstructionLoss.forward (losses.py:73-80)."""
    yf = (y * mask).reshape(y.shape[0], -1)
    yhf = (yh * mask).reshape(yh.shape[0], -1)
    return (cfg.l1 * F.l1_loss(yf, yhf).mean(0).sum()
            + cfg.l2 * F.mse_loss(yf, yhf).mean(0).sum()
            + cfg.linf * torch.topk((yf - yhf) ** 2, cfg.linf_topk, dim=-1)[0].mean(0).sum())


# ---------------------------------------------------------------------------
# Whole model (models/vqvae/vqvae.py) and one train step (train.py:82-143)
# ---------------------------------------------------------------------------

def vqvae_forward(x: Tensor, x_lengths: Tensor, p: Params, cfg: VQVAEConfig, state: CodebookState,
                  training: bool, drop: DropFn = no_dropout, k_rand: Optional[Tensor] = None,
                  exact_indices: bool = True, reduce_fn=None):
    """VQVAE.forward (vqvae.py:98-132).  Returns (loss_dict, metrics, aux)."""
    x_mask = sequence_mask(x_lengths, x.size(2)).unsqueeze(1).to(x.dtype)
    z, z_mask = encoder_forward(x, x_mask, p, cfg, drop if training else no_dropout)
    x_l, xq, commit, metrics = vq_forward(z, z_mask, state, cfg.mu, cfg.revival_threshold,
                                          update_k=training, k_rand=k_rand,
                                          exact_indices=exact_indices, reduce_fn=reduce_fn)
    if not training:
        xq = xq.detach()  # bottleneck.py:230-233
    x_out, _ = decoder_forward(xq, z_mask, p, cfg, drop if training else no_dropout)
    assert x_out.shape == x.shape
    loss_recon = multinorm_recon_loss(x, x_out, x_mask, cfg)
    loss_stft = multires_stft_loss(x, x_out, x_mask, cfg)
    loss = loss_recon + cfg.multispectral * loss_stft + cfg.commit * commit
    out = {"loss": loss, "loss_recon": loss_recon, "loss_stft": loss_stft, "loss_commit": commit,
           "yh": x_out.squeeze(1)}
    return out, (metrics if training else {}), {"z": z, "z_mask": z_mask, "codes": x_l}


def param_shapes(cfg: VQVAEConfig) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of the trainable parameters in reference order
    (encoders.0..., decoders.0...) for the effective last level."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    w, e = cfg.eff_width, cfg.emb_width

    def block(prefix):
        for d in range(cfg.eff_depth):
            k, _, _ = branch_geometry(cfg, d)
            bp = f"{prefix}.blocks.{d}"
            shapes[bp + ".0.weight"] = (2 * w, w, 1)
            shapes[bp + ".0.bias"] = (2 * w,)
            shapes[bp + ".1.model.2.weight"] = (2 * w, 2 * w, k)
            shapes[bp + ".1.model.2.bias"] = (2 * w,)
            shapes[bp + ".1.model.5.weight"] = (2 * w, 2 * w, 1)
            shapes[bp + ".1.model.5.bias"] = (2 * w,)
        shapes[prefix + ".gate.weight"] = (w, w, 1)
        shapes[prefix + ".gate.bias"] = (w,)

    for level in range(cfg.levels):
        lp = f"encoders.0.level_blocks.{level}.blocks"
        s = cfg.strides_t[level]
        i = 0
        for j in range(cfg.downs_t[level]):
            cin = (1 if level == 0 else e) if j == 0 else w
            shapes[f"{lp}.{i}.weight"] = (w, cin, 2 * s)
            shapes[f"{lp}.{i}.bias"] = (w,)
            block(f"{lp}.{i + 1}")
            i += 2
        shapes[f"{lp}.{i}.weight"] = (e, w, 3)
        shapes[f"{lp}.{i}.bias"] = (e,)
    for level in range(cfg.levels):
        lp = f"decoders.0.level_blocks.{level}.blocks"
        s = cfg.strides_t[level]
        shapes[f"{lp}.0.weight"] = (w, e, 3)
        shapes[f"{lp}.0.bias"] = (w,)
        i = 1
        for j in range(cfg.downs_t[level]):
            block(f"{lp}.{i}")
            cout = e if j == cfg.downs_t[level] - 1 else w
            shapes[f"{lp}.{i + 1}.weight"] = (w, cout, 2 * s)  # ConvTranspose1d: [Cin, Cout, k]
            shapes[f"{lp}.{i + 1}.bias"] = (cout,)
            i += 2
    shapes["decoders.0.out.weight"] = (1, e, 1)
    shapes["decoders.0.out.bias"] = (1,)
    return shapes


def init_params(cfg: VQVAEConfig, seed: int = 0, zero_out: Optional[bool] = None) -> Params:
    """Random parameters with torch's default Conv1d init bounds
    (kaiming_uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)) for weight and bias) and
    the reference's zero-init of ResLayer's last conv and the gate
    (resnet.py:29-32, :218-220).  Values are NOT the reference's RNG stream;
    parity tests load captured state dicts instead."""
    g = torch.Generator().manual_seed(seed)
    zero_out = cfg.zero_out if zero_out is None else zero_out
    p: Params = {}
    shapes = param_shapes(cfg)
    for name, shape in shapes.items():
        if not name.endswith(".weight"):
            continue
        # torch derives fan_in from weight.size(1) * k for Conv1d AND ConvTranspose1d
        bound = 1.0 / math.sqrt(shape[1] * shape[2])
        p[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        bname = name[:-6] + "bias"
        p[bname] = (torch.rand(shapes[bname], generator=g) * 2 - 1) * bound
    p = {name: p[name] for name in shapes}
    if zero_out:
        for name in p:
            if ".model.5." in name or ".gate." in name:
                p[name] = torch.zeros_like(p[name])
    return p


# ---------------------------------------------------------------------------
# Counter-based dropout masks: the PRODUCT's generator restated bit-exactly.
# (The reference draws from torch's global RNG, resnet.py:22,25, which no device
# can reproduce; the build defines its own stateless generator so that train
# mode is parity-testable.  Spec: include/smt_hip.h 'dropout'.)
# ---------------------------------------------------------------------------

def _fmix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def dropout_site_key(seed: int, site: int) -> int:
    """32-bit per-site key: fmix32(seed * 0x9E3779B1 + site * 0x7F4A7C15 + 1)."""
    v = (seed * 0x9E3779B1 + site * 0x7F4A7C15 + 1) & 0xFFFFFFFF
    return int(_fmix32(np.array([v], dtype=np.uint32))[0])


def dropout_keep_ntc(seed: int, site: int, b: int, t: int, c: int, p: float) -> np.ndarray:
    """keep[b, t, c] (bool) for linear channels-last index i = (b*T + t)*C + c (include/smt_hip.h
    "dropout"): 16 random bits per element, two elements per hash:
        h = fmix32((uint32)(i >> 1) * 0x9E3779B1 + key);  keep iff ((h >> 16*(i&1)) & 0xFFFF) >= round(p*65536)."""
    key = np.uint32(dropout_site_key(seed, site))
    i = np.arange(b * t * c, dtype=np.uint64)
    h = (((i >> np.uint64(1)) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    h = _fmix32((h + key).astype(np.uint32))
    bits = (h >> (np.uint32(16) * (i & np.uint64(1)).astype(np.uint32))) & np.uint32(0xFFFF)
    thresh = np.uint32(int(round(p * 65536.0)))
    return (bits >= thresh).reshape(b, t, c)


def make_counter_dropout(seed: int, p: float, site_ids: Dict[str, int]) -> DropFn:
    """DropFn applying the counter-based mask; ``x`` is NCT so the NTC-indexed
    mask is transposed.  Scale 1/(1-p) in fp32."""
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))

    def drop(site: str, x: Tensor) -> Tensor:
        b, c, t = x.shape
        keep = dropout_keep_ntc(seed, site_ids[site], b, t, c, p)
        m = torch.from_numpy(keep.transpose(0, 2, 1).astype(np.float32) * scale)
        return x * m

    return drop


def dropout_site_ids(cfg: VQVAEConfig) -> Dict[str, int]:
    """Deterministic numbering of the 2*depth dropout sites of every GatedHiFi
    block in forward order (encoder blocks first, then decoder blocks)."""
    ids: Dict[str, int] = {}
    n = 0
    for level in range(cfg.levels):
        lp = f"encoders.0.level_blocks.{level}.blocks"
        for j in range(cfg.downs_t[level]):
            for d in range(cfg.eff_depth):
                for s in ("drop0", "drop1"):
                    ids[f"{lp}.{2 * j + 1}.blocks.{d}.1.{s}"] = n
                    n += 1
    for level in reversed(range(cfg.levels)):
        lp = f"decoders.0.level_blocks.{level}.blocks"
        for j in range(cfg.downs_t[level]):
            for d in range(cfg.eff_depth):
                for s in ("drop0", "drop1"):
                    ids[f"{lp}.{2 * j + 1}.blocks.{d}.1.{s}"] = n
                    n += 1
    return ids


# ---------------------------------------------------------------------------
# Synthetic LJSpeech-shaped clips; shared spec with the product's
# generator, restated here so the CPU baseline sees the same batch.
# ---------------------------------------------------------------------------

def synthetic_clip_batch(batch: int, length: int, seed: int, sample_rate: int = 22050) -> Tensor:
    """0.5*(0.6*sum_{h=1..8} sin(2 pi h f0 t + phi_h)/h + 0.4*U(-1,1)), f0~U[90,250],
    clamped to [-1, 1]; torch.Generator().manual_seed(seed).  Returns [B,1,T] f32."""
    g = torch.Generator().manual_seed(seed)
    f0 = 90.0 + 160.0 * torch.rand(batch, 1, generator=g, dtype=torch.float64)
    phi = 2 * math.pi * torch.rand(batch, 8, generator=g, dtype=torch.float64)
    t = torch.arange(length, dtype=torch.float64)[None, :] / sample_rate
    tone = torch.zeros(batch, length, dtype=torch.float64)
    for h in range(1, 9):
        tone += torch.sin(2 * math.pi * h * f0 * t + phi[:, h - 1:h]) / h
    noise = torch.rand(batch, length, generator=g, dtype=torch.float64) * 2 - 1
    x = 0.5 * (0.6 * tone + 0.4 * noise)
    return x.clamp(-1, 1).to(torch.float32).unsqueeze(1)


# ---------------------------------------------------------------------------
# CPU train step (train.py:82-143, full-precision branch) -- used as the
# ``cpu_baseline`` in bench.py.
# ---------------------------------------------------------------------------

class OracleTrainer:
    """AdamW(lr 1e-4, betas (0.9, 0.98), eps 1e-9, wd 0) + constant LR
    (utils/commons.py:126-134, configs/models/vqvae.yaml:42-49) around
    ``vqvae_forward``.  Dropout uses torch's CPU generator like the reference."""

    def __init__(self, cfg: VQVAEConfig, seed: int = 0):
        self.cfg = cfg
        self.params = {n: v.clone().requires_grad_(True) for n, v in init_params(cfg, seed).items()}
        self.state = CodebookState(k=torch.zeros(cfg.l_bins, cfg.emb_width))
        self.opt = torch.optim.AdamW(list(self.params.values()), lr=1e-4, betas=(0.9, 0.98),
                                     eps=1e-9, weight_decay=0)
        self.gen = torch.Generator().manual_seed(seed + 1)

    def _drop(self, site: str, x: Tensor) -> Tensor:
        return F.dropout(x, p=self.cfg.dropout, training=True)

    def _k_rand(self, rows: Tensor) -> Tensor:
        rows = vq_tile(rows, self.cfg.l_bins)
        return rows[torch.randperm(rows.shape[0], generator=self.gen)][:self.cfg.l_bins]

    def step(self, x: Tensor, x_lengths: Tensor) -> Dict[str, float]:
        self.opt.zero_grad()
        out, _, _ = vqvae_forward(x, x_lengths, self.params, self.cfg, self.state, True,
                                  drop=self._drop, k_rand=self._k_rand, exact_indices=False)
        if torch.isnan(out["loss"]):
            raise RuntimeError("NaN loss")  # train.py:124-133
        out["loss"].backward()
        self.opt.step()
        return {k: float(v) for k, v in out.items() if k.startswith("loss")}
