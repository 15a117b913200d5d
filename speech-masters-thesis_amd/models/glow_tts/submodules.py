"""Building blocks of GlowTTS (reference models/glow_tts/submodules.py) on channels-last activations [B, T, C] with prefix row
masks carried as lengths.  Module and parameter names follow the reference, so its checkpoints load unchanged (weight-normed
convolutions keep ``weight_g`` / ``weight_v``).  Arithmetic: convolutions on the MFMA implicit-GEMM kernels
(``smt_amd.convops``), LayerNorm / ReLU + dropout on the kernels of ``smt_amd.lm``, everything else on ``smt_amd.glow``
(csrc/glow.hip); the monotonic alignment search is ``smt_maximum_path`` (csrc/mas.hip).  No op here has a torch fallback."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from smt_amd import convops, glow
from smt_amd import lm as K
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


def generate_path(duration, mask):
    """Alignment from predicted durations (submodules.py:70-85): duration [b, t_x], mask [b, t_x, t_y]."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1)
    path = sequence_mask(cum.view(b * t_x), t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0))[:, :-1]
    return path * mask


# ---- convolution helpers ---------------------------------------------------------------------------------------------
def _cin_ok(c):
    """Input-channel counts the fp32 implicit-GEMM kernel takes (include/smt_hip.h, smt_conv1d_ntc)."""
    return c % 8 == 0 and ((c <= 64 and (c & (c - 1)) == 0) or c % 64 == 0)


def conv(x, weight, bias, *, padding=0, dilation=1, lens=None, residual=None, x_channels=None):
    """Conv1d on channels-last rows.  ``x`` may be wider than the layer's input (``x_channels`` = how many leading channels the
    layer reads): channel counts the MFMA kernel does not take (80 mels) are run as the next admissible count with zero weights
    on the extra channels -- exact, and the rows are already in memory.  Output channel counts are padded the same way (zero
    weight rows; the extra output channels are sliced off), because the data gradient reads them as ITS input channels."""
    c_out, c_in, _ = weight.shape
    if x_channels is None:
        x_channels = x.shape[-1]
    assert x_channels == c_in
    c_use = c_in
    while not _cin_ok(c_use):
        c_use += 8
    assert c_use <= x.shape[-1], f"{c_in} input channels need padding to {c_use}, the rows have {x.shape[-1]}"
    if c_use != c_in:
        weight = F.pad(weight, (0, 0, 0, c_use - c_in))
    o_use = c_out                      # the data gradient is a convolution over the OUTPUT channels: same admissible counts
    while not _cin_ok(o_use):
        o_use += 8 - o_use % 8 if o_use % 8 else 8
    if o_use != c_out:
        weight = F.pad(weight, (0, 0, 0, 0, 0, o_use - c_out))
        bias = F.pad(bias, (0, o_use - c_out)) if bias is not None else None
    y = convops.conv1d(x[:, :, :c_use], weight, bias, padding=padding, dilation=dilation, lens=lens, residual=residual)
    return y if o_use == c_out else y[:, :, :c_out]


class ConvParams(nn.Module):
    """weight / bias of one nn.Conv1d (torch's layout and default init)."""

    def __init__(self, c_in, c_out, kernel, zero=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in, kernel))
        self.bias = nn.Parameter(torch.empty(c_out))
        bound = 1.0 / math.sqrt(c_in * kernel)
        with torch.no_grad():
            if zero:
                self.weight.zero_(); self.bias.zero_()
            else:
                self.weight.uniform_(-bound, bound); self.bias.uniform_(-bound, bound)


class WeightNormConvParams(nn.Module):
    """torch.nn.utils.weight_norm(Conv1d): parameters ``weight_g`` [c_out, 1, 1], ``weight_v`` [c_out, c_in, k], ``bias``."""

    def __init__(self, c_in, c_out, kernel):
        super().__init__()
        bound = 1.0 / math.sqrt(c_in * kernel)
        v = torch.empty(c_out, c_in, kernel).uniform_(-bound, bound)
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.empty(c_out).uniform_(-bound, bound))

    @property
    def weight(self):
        v = self.weight_v
        return self.weight_g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)


class LayerNorm(nn.Module):
    """Channel LayerNorm (submodules.py:98-116), eps 1e-4; ``forward(x, h, drop)`` = LN(x + dropout(h))."""

    def __init__(self, channels, eps=1e-4):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x, h=None, drop=K.NO_DROP):
        return K.add_layer_norm(x, h, self.gamma, self.beta, eps=self.eps, drop=drop)


class _Sites:
    """Dropout site numbering of one model: names in forward order -> ids (the oracle replays the same masks by name)."""

    def __init__(self):
        self.names = []

    def add(self, name):
        self.names.append(name)
        return len(self.names) - 1


class ConvReluNorm(nn.Module):
    """submodules.py:119-164: n x [conv k, LayerNorm, ReLU, dropout], zero-initialised 1x1 projection, residual."""

    def __init__(self, in_channels, hidden_channels, out_channels, kernel_size, n_layers, p_dropout, sites, prefix):
        super().__init__()
        assert n_layers > 1
        self.kernel_size, self.n_layers, self.p_dropout = kernel_size, n_layers, p_dropout
        self.conv_layers = nn.ModuleList(ConvParams(in_channels if i == 0 else hidden_channels, hidden_channels, kernel_size)
                                         for i in range(n_layers))
        self.norm_layers = nn.ModuleList(LayerNorm(hidden_channels) for _ in range(n_layers))
        self.proj = ConvParams(hidden_channels, out_channels, 1, zero=True)
        self.sites = [sites.add(f"{prefix}.relu_drop.{i}") for i in range(n_layers)]
        self._zero = None

    def forward(self, x, lens, seed):
        x_org = x
        for i in range(self.n_layers):
            c = self.conv_layers[i]
            x = conv(x, c.weight, c.bias, padding=self.kernel_size // 2, lens=lens)
            x = self.norm_layers[i](x)
            if self._zero is None or self._zero.device != x.device or self._zero.numel() != x.shape[-1]:
                self._zero = torch.zeros(x.shape[-1], device=x.device)
            x = K.bias_relu_dropout_(x, self._zero, K.Drop(self.p_dropout, self.training, seed, self.sites[i]))
        return conv(x, self.proj.weight, self.proj.bias, residual=x_org)        # `* x_mask`: the consumers mask their input rows


class WN(nn.Module):
    """submodules.py:167-228 without speaker conditioning: n layers of [dilated conv H -> 2H, dropout, tanh * sigmoid gate,
    1x1 conv to (residual | skip)].  The residual and skip halves of ``res_skip_layers`` run as two convolutions with the
    running tensors as their residual operands, so `x + res` and `output + skip` cost no extra pass."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout, sites, prefix):
        super().__init__()
        assert kernel_size % 2 == 1 and hidden_channels % 2 == 0
        self.hidden_channels, self.kernel_size, self.dilation_rate, self.n_layers = hidden_channels, kernel_size, dilation_rate, n_layers
        self.p_dropout = p_dropout
        self.in_layers = nn.ModuleList(WeightNormConvParams(hidden_channels, 2 * hidden_channels, kernel_size) for _ in range(n_layers))
        self.res_skip_layers = nn.ModuleList(
            WeightNormConvParams(hidden_channels, 2 * hidden_channels if i < n_layers - 1 else hidden_channels, 1) for i in range(n_layers))
        self.sites = [sites.add(f"{prefix}.drop.{i}") for i in range(n_layers)]

    def forward(self, x, lens, seed):
        h, output = self.hidden_channels, None
        for i in range(self.n_layers):
            dil = self.dilation_rate ** i
            il, rs = self.in_layers[i], self.res_skip_layers[i]
            a = conv(x, il.weight, il.bias, padding=(self.kernel_size * dil - dil) // 2, dilation=dil, lens=lens)
            acts = glow.wn_gate(a, K.Drop(self.p_dropout, self.training, seed, self.sites[i]))
            w, b = rs.weight, rs.bias
            if i < self.n_layers - 1:
                x = conv(acts, w[:h], b[:h], residual=x)                   # rows beyond the length are masked by every reader
                output = conv(acts, w[h:], b[h:], residual=output)
            else:
                output = conv(acts, w, b, residual=output)
        return output


class ActNorm(nn.Module):
    """submodules.py:231-274."""

    def __init__(self, channels, ddi=False, **kwargs):
        super().__init__()
        self.channels, self.initialized = channels, not ddi
        self.logs = nn.Parameter(torch.zeros(1, channels, 1))
        self.bias = nn.Parameter(torch.zeros(1, channels, 1))

    def forward(self, x, lens, reverse=False, **kwargs):
        if not self.initialized:
            self.initialize(x, lens)
            self.initialized = True
        if reverse:
            return glow.actnorm_reverse(x, self.logs, self.bias, lens), None
        z = glow.actnorm(x, self.logs, self.bias, lens)
        return z, torch.sum(self.logs) * lens.to(x.dtype)

    def store_inverse(self):
        pass

    def set_ddi(self, ddi):
        self.initialized = not ddi

    @torch.no_grad()
    def initialize(self, x, lens):
        cnt, s1, s2 = glow.masked_channel_moments(x, lens)
        m, m_sq = s1 / cnt, s2 / cnt
        logs = 0.5 * torch.log(torch.clamp_min(m_sq - m ** 2, 1e-6))
        self.bias.data.copy_((-m * torch.exp(-logs)).view_as(self.bias))
        self.logs.data.copy_((-logs).view_as(self.logs))


class InvConvNear(nn.Module):
    """submodules.py:277-326."""

    def __init__(self, channels, n_split=4, no_jacobian=False, **kwargs):
        super().__init__()
        assert n_split == 4, "csrc/glow.hip mixes channel quadruples (n_split = 4, the reference's configuration)"
        self.channels, self.n_split, self.no_jacobian = channels, n_split, no_jacobian
        w = torch.linalg.qr(torch.empty(n_split, n_split).normal_())[0]
        if torch.det(w) < 0:
            w[:, 0] = -w[:, 0]
        self.weight = nn.Parameter(w.contiguous())

    def forward(self, x, lens, reverse=False, **kwargs):
        c = x.shape[-1]
        if reverse:
            w_inv = self.weight_inv if hasattr(self, "weight_inv") else torch.inverse(self.weight.float())
            return glow.invconv_reverse(x, w_inv.contiguous(), lens), None
        z = glow.invconv(x, self.weight, lens)
        logdet = 0 if self.no_jacobian else torch.logdet(self.weight) * (c / self.n_split) * lens.to(x.dtype)
        return z, logdet

    def store_inverse(self):
        self.weight_inv = torch.inverse(self.weight.float()).to(dtype=self.weight.dtype)


class CouplingBlock(nn.Module):
    """submodules.py:329-408."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0, sigmoid_scale=False,
                 sites=None, prefix=""):
        super().__init__()
        assert gin_channels == 0, "speaker conditioning is not built (n_speakers = 1 in configs/models/glow_tts.yaml)"
        self.in_channels, self.hidden_channels, self.sigmoid_scale = in_channels, hidden_channels, sigmoid_scale
        self.start = WeightNormConvParams(in_channels // 2, hidden_channels, 1)
        self.end = ConvParams(hidden_channels, in_channels, 1, zero=True)    # zero: the coupling starts as the identity
        self.wn = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout, sites if sites is not None else _Sites(), prefix + ".wn")

    def forward(self, x, lens, reverse=False, seed=0, **kwargs):
        half = self.in_channels // 2
        h = conv(x, self.start.weight, self.start.bias, lens=lens, x_channels=half)       # reads x_0 = the first half of the row
        h = self.wn(h, lens, seed)
        out = conv(h, self.end.weight, self.end.bias, lens=lens)
        if reverse:
            return glow.coupling_reverse(out, x, lens, self.sigmoid_scale), None
        return glow.coupling(out, x, lens, self.sigmoid_scale)

    def store_inverse(self):
        pass


class AttentionBlock(nn.Module):
    """submodules.py:411-575: self-attention with relative-position keys / values (window_size, heads_share = True)."""

    def __init__(self, channels, out_channels, n_heads, window_size=None, heads_share=True, p_dropout=0.0, block_length=None,
                 proximal_bias=False, proximal_init=False, sites=None, prefix=""):
        super().__init__()
        assert channels % n_heads == 0 and window_size is not None and heads_share and block_length is None and not proximal_bias
        self.channels, self.out_channels, self.n_heads, self.window_size, self.p_dropout = channels, out_channels, n_heads, window_size, p_dropout
        self.k_channels = channels // n_heads
        self.conv_q, self.conv_k, self.conv_v = (ConvParams(channels, channels, 1) for _ in range(3))
        self.conv_o = ConvParams(channels, out_channels, 1)
        std = self.k_channels ** -0.5
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * std)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * std)
        for c in (self.conv_q, self.conv_k, self.conv_v):
            nn.init.xavier_uniform_(c.weight)
        if proximal_init:
            self.conv_k.weight.data.copy_(self.conv_q.weight.data); self.conv_k.bias.data.copy_(self.conv_q.bias.data)
        self.site = (sites if sites is not None else _Sites()).add(prefix + ".drop")

    def forward(self, x, lens, seed):
        q = conv(x, self.conv_q.weight, self.conv_q.bias, lens=lens)
        k = conv(x, self.conv_k.weight, self.conv_k.bias, lens=lens)
        v = conv(x, self.conv_v.weight, self.conv_v.bias, lens=lens)
        ctx = glow.rel_attention(q, k, v, self.emb_rel_k, self.emb_rel_v, lens, self.n_heads, self.window_size,
                                 K.Drop(self.p_dropout, self.training, seed, self.site))
        return conv(ctx, self.conv_o.weight, self.conv_o.bias)


class FeedForwardNetwork(nn.Module):
    """submodules.py:578-609 (relu activation)."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0, activation=None, sites=None, prefix=""):
        super().__init__()
        assert activation is None, "only the relu feed-forward of the reference configuration is built"
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        self.conv_1 = ConvParams(in_channels, filter_channels, kernel_size)
        self.conv_2 = ConvParams(filter_channels, out_channels, kernel_size)
        self.site = (sites if sites is not None else _Sites()).add(prefix + ".drop")

    def forward(self, x, lens, seed):
        pad = self.kernel_size // 2
        h = conv(x, self.conv_1.weight, None, padding=pad, lens=lens)
        h = K.bias_relu_dropout_(h, self.conv_1.bias, K.Drop(self.p_dropout, self.training, seed, self.site))
        return conv(h, self.conv_2.weight, self.conv_2.bias, padding=pad, lens=lens)


class DurationPredictor(nn.Module):
    """submodules.py:612-637: 2 x [conv k, ReLU, LayerNorm, dropout], 1x1 projection to one channel."""

    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, sites=None, prefix=""):
        super().__init__()
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        self.conv_1 = ConvParams(in_channels, filter_channels, kernel_size)
        self.norm_1 = LayerNorm(filter_channels)
        self.conv_2 = ConvParams(filter_channels, filter_channels, kernel_size)
        self.norm_2 = LayerNorm(filter_channels)
        self.proj = ConvParams(filter_channels, 1, 1)
        s = sites if sites is not None else _Sites()
        self.sites = [s.add(prefix + ".drop.0"), s.add(prefix + ".drop.1")]

    def forward(self, x, lens, seed):
        pad = self.kernel_size // 2
        for i, (c, n) in enumerate(((self.conv_1, self.norm_1), (self.conv_2, self.norm_2))):
            h = conv(x, c.weight, None, padding=pad, lens=lens)
            h = K.bias_relu_dropout_(h, c.bias)                                  # relu(conv + bias)
            x = glow.dropout(n(h), K.Drop(self.p_dropout, self.training, seed, self.sites[i]))
        return conv(x, self.proj.weight, self.proj.bias, lens=lens)[:, :, 0]     # [B, T]; rows beyond the length: masked by the loss
