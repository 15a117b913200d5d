"""GlowTTS ops on the HIP library (include/smt_hip.h, "GlowTTS"): every function is a torch.autograd.Function whose forward
and backward are libsmt_hip.so launches on torch's current stream.  fp32, channels-last rows [B, T, C], prefix row masks as
int32 lengths.  Reference: models/glow_tts/{glow_tts,modules,submodules}.py (cited per function)."""
import math

import torch

from . import native as N
from . import profiler
from .lm import NO_DROP


def _f(t):
    assert t.dtype == torch.float32 and t.is_cuda, "GlowTTS kernels take fp32 device tensors"
    return t.contiguous()


def _lens(lens):
    if lens is None:
        return None
    assert lens.is_cuda
    return lens if lens.dtype == torch.int32 else lens.to(torch.int32)


def _ws(nbytes, device):
    return N.workspace.get(max(int(nbytes), 16), device)


class _ActNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, logs, bias, lens):
        x = _f(x)
        b, t, c = x.shape
        z = torch.empty_like(x)
        lg, bs = _f(logs.reshape(-1)), _f(bias.reshape(-1))
        with profiler.region("glow_actnorm:fwd", nbytes=2 * x.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_glow_actnorm_fwd(N.ptr(x), N.ptr(lg), N.ptr(bs), N.ptr(lens), N.ptr(z), b, t, c, 0, N.stream_ptr()),
                    "smt_glow_actnorm_fwd")
        ctx.save_for_backward(x, lg, lens if lens is not None else torch.empty(0))
        ctx.meta = (logs.shape, bias.shape, lens is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, lg, lens = ctx.saved_tensors
        logs_shape, bias_shape, has_lens = ctx.meta
        lens = lens if has_lens else None
        b, t, c = x.shape
        dz = _f(dz)
        dx = torch.empty_like(x)
        dlogs, dbias = torch.empty(c, device=x.device), torch.empty(c, device=x.device)
        lib = N.lib()
        ws = _ws(lib.smt_glow_reduce_workspace_bytes(b * t, 2 * c), x.device)
        with profiler.region("glow_actnorm:bwd", nbytes=3 * x.numel() * 4, bound="hbm"):
            N.check(lib.smt_glow_actnorm_bwd(N.ptr(x), N.ptr(dz), N.ptr(lg), N.ptr(lens), N.ptr(dx), N.ptr(dlogs), N.ptr(dbias), b, t, c,
                                             N.ptr(ws), ws.numel(), N.stream_ptr()), "smt_glow_actnorm_bwd")
        return dx, dlogs.view(logs_shape), dbias.view(bias_shape), None


def actnorm(x, logs, bias, lens):
    """z = (bias + exp(logs) x) mask (ActNorm.forward, submodules.py:237-253); the log-determinant sum(logs) * len is a
    product of two small tensors and is formed by the caller."""
    return _ActNorm.apply(x, logs, bias, _lens(lens))


@torch.no_grad()
def actnorm_reverse(x, logs, bias, lens):
    x = _f(x)
    b, t, c = x.shape
    z = torch.empty_like(x)
    N.check(N.lib().smt_glow_actnorm_fwd(N.ptr(x), N.ptr(_f(logs.reshape(-1))), N.ptr(_f(bias.reshape(-1))), N.ptr(_lens(lens)), N.ptr(z),
                                         b, t, c, 1, N.stream_ptr()), "smt_glow_actnorm_fwd")
    return z


@torch.no_grad()
def masked_channel_moments(x, lens):
    """(count, sum x mask, sum x^2 mask) per channel -- the statistics of ActNorm.initialize (submodules.py:261-274), from two
    calls of the ActNorm backward kernel, whose column sums are exactly these (dz = 1: sum mask and sum x mask; dz = x: sum x^2 mask)."""
    x = _f(x)
    b, t, c = x.shape
    lib = N.lib()
    zeros = torch.zeros(c, device=x.device)
    ws = _ws(lib.smt_glow_reduce_workspace_bytes(b * t, 2 * c), x.device)
    s1, cnt, s2, tmp = (torch.empty(c, device=x.device) for _ in range(4))
    ones = torch.ones_like(x)
    N.check(lib.smt_glow_actnorm_bwd(N.ptr(x), N.ptr(ones), N.ptr(zeros), N.ptr(_lens(lens)), None, N.ptr(s1), N.ptr(cnt), b, t, c, N.ptr(ws),
                                     ws.numel(), N.stream_ptr()), "smt_glow_actnorm_bwd")
    N.check(lib.smt_glow_actnorm_bwd(N.ptr(x), N.ptr(x), N.ptr(zeros), N.ptr(_lens(lens)), None, N.ptr(s2), N.ptr(tmp), b, t, c, N.ptr(ws),
                                     ws.numel(), N.stream_ptr()), "smt_glow_actnorm_bwd")
    return cnt, s1, s2


class _InvConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, lens):
        x, w = _f(x), _f(weight)
        b, t, c = x.shape
        z = torch.empty_like(x)
        with profiler.region("glow_invconv:fwd", nbytes=2 * x.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_glow_invconv(N.ptr(x), N.ptr(w), N.ptr(lens), N.ptr(z), b, t, c, 0, N.stream_ptr()), "smt_glow_invconv")
        ctx.save_for_backward(x, w, lens if lens is not None else torch.empty(0))
        ctx.has_lens = lens is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, lens = ctx.saved_tensors
        lens = lens if ctx.has_lens else None
        b, t, c = x.shape
        dz = _f(dz)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        lib = N.lib()
        ws = _ws(lib.smt_glow_reduce_workspace_bytes(b * t, 16), x.device)
        with profiler.region("glow_invconv:bwd", nbytes=4 * x.numel() * 4, bound="hbm"):
            N.check(lib.smt_glow_invconv(N.ptr(dz), N.ptr(w), N.ptr(lens), N.ptr(dx), b, t, c, 1, N.stream_ptr()), "smt_glow_invconv")
            N.check(lib.smt_glow_invconv_wgrad(N.ptr(x), N.ptr(dz), N.ptr(lens), N.ptr(dw), b, t, c, N.ptr(ws), ws.numel(), N.stream_ptr()),
                    "smt_glow_invconv_wgrad")
        return dx, dw, None


def invconv(x, weight, lens):
    """InvConvNear.forward (submodules.py:292-323), n_split = 4: z = (W applied to channel quadruples) mask."""
    assert weight.shape == (4, 4), "the HIP kernel is built for n_split = 4 (configs/models/glow_tts.yaml)"
    return _InvConv.apply(x, weight, _lens(lens))


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, drop):
        a = _f(a)
        h = a.shape[-1] // 2
        rows = a.numel() // (2 * h)
        acts = torch.empty(*a.shape[:-1], h, device=a.device)
        with profiler.region("glow_gate:fwd", nbytes=3 * rows * h * 4, bound="hbm"):
            N.check(N.lib().smt_glow_gate_fwd(N.ptr(a), N.ptr(acts), rows, h, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale,
                                              N.stream_ptr()), "smt_glow_gate_fwd")
        ctx.save_for_backward(a)
        ctx.drop = drop
        return acts

    @staticmethod
    def backward(ctx, dacts):
        (a,) = ctx.saved_tensors
        drop = ctx.drop
        h = a.shape[-1] // 2
        rows = a.numel() // (2 * h)
        da = torch.empty_like(a)
        with profiler.region("glow_gate:bwd", nbytes=5 * rows * h * 4, bound="hbm"):
            N.check(N.lib().smt_glow_gate_bwd(N.ptr(a), N.ptr(_f(dacts)), N.ptr(da), rows, h, drop.key, N.ptr(drop.key_dev), drop.thresh,
                                              drop.scale, N.stream_ptr()), "smt_glow_gate_bwd")
        return da, None


def wn_gate(a, drop=NO_DROP):
    """tanh(d(a)[..., :H]) * sigmoid(d(a)[..., H:]) with d = the in_layer's dropout (WN.forward, submodules.py:213-220)."""
    return _Gate.apply(a, drop)


class _Coupling(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, x, lens, sigmoid_scale):
        out, x = _f(out), _f(x)
        b, t, c = x.shape
        z = torch.empty_like(x)
        logdet = torch.empty(b, device=x.device)
        ws = _ws(4 * b * ((t + 63) // 64), x.device)
        with profiler.region("glow_coupling:fwd", nbytes=3 * x.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_glow_coupling_fwd(N.ptr(out), N.ptr(x), N.ptr(lens), N.ptr(z), N.ptr(logdet), b, t, c, int(sigmoid_scale), 0,
                                                  N.ptr(ws), ws.numel(), N.stream_ptr()), "smt_glow_coupling_fwd")
        ctx.save_for_backward(out, x, lens if lens is not None else torch.empty(0))
        ctx.meta = (lens is not None, int(sigmoid_scale))
        return z, logdet

    @staticmethod
    def backward(ctx, dz, dlogdet):
        out, x, lens = ctx.saved_tensors
        has_lens, sig = ctx.meta
        lens = lens if has_lens else None
        b, t, c = x.shape
        dout, dx = torch.empty_like(out), torch.empty_like(x)
        dld = None if dlogdet is None else _f(dlogdet)
        with profiler.region("glow_coupling:bwd", nbytes=5 * x.numel() * 4, bound="hbm"):
            N.check(N.lib().smt_glow_coupling_bwd(N.ptr(out), N.ptr(x), N.ptr(_f(dz)), N.ptr(dld), N.ptr(lens), N.ptr(dout), N.ptr(dx), b, t, c,
                                                  sig, N.stream_ptr()), "smt_glow_coupling_bwd")
        return dout, dx, None, None


def coupling(out, x, lens, sigmoid_scale=False):
    """(z, logdet[b]) of the affine coupling (CouplingBlock.forward, submodules.py:392-405): out = (m | logs), x = (x0 | x1)."""
    return _Coupling.apply(out, x, _lens(lens), sigmoid_scale)


@torch.no_grad()
def coupling_reverse(out, x, lens, sigmoid_scale=False):
    out, x = _f(out), _f(x)
    b, t, c = x.shape
    z = torch.empty_like(x)
    N.check(N.lib().smt_glow_coupling_fwd(N.ptr(out), N.ptr(x), N.ptr(_lens(lens)), N.ptr(z), None, b, t, c, int(sigmoid_scale), 1, None, 0,
                                          N.stream_ptr()), "smt_glow_coupling_fwd")
    return z


@torch.no_grad()
def invconv_reverse(x, weight_inv, lens):
    x = _f(x)
    b, t, c = x.shape
    z = torch.empty_like(x)
    N.check(N.lib().smt_glow_invconv(N.ptr(x), N.ptr(_f(weight_inv)), N.ptr(_lens(lens)), N.ptr(z), b, t, c, 0, N.stream_ptr()), "smt_glow_invconv")
    return z


class _RelAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, ek, ev, lens, heads, window, drop):
        q, k, v = _f(q), _f(k), _f(v)
        b, t, c = q.shape
        d = c // heads
        ek2, ev2 = _f(ek.reshape(2 * window + 1, d)), _f(ev.reshape(2 * window + 1, d))
        ctxv = torch.empty_like(q)
        probs = torch.empty(b, heads, t, t, device=q.device)
        with profiler.region("glow_attention:fwd", flops=4.0 * b * heads * t * t * d, bound="mfma", dtype="f32"):
            N.check(N.lib().smt_glow_attention_fwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(ek2), N.ptr(ev2), N.ptr(lens), N.ptr(ctxv), N.ptr(probs), b,
                                                   t, heads, d, window, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale,
                                                   N.stream_ptr()), "smt_glow_attention_fwd")
        ctx.save_for_backward(q, k, v, ek2, ev2, probs)
        ctx.meta = (heads, window, drop, ek.shape, ev.shape)
        return ctxv

    @staticmethod
    def backward(ctx, dctx):
        q, k, v, ek2, ev2, probs = ctx.saved_tensors
        heads, window, drop, ek_shape, ev_shape = ctx.meta
        b, t, c = q.shape
        d = c // heads
        lib = N.lib()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dek, dev = torch.empty_like(ek2), torch.empty_like(ev2)
        ws = _ws(lib.smt_glow_attention_bwd_workspace_bytes(b, t, heads, d, window), q.device)
        with profiler.region("glow_attention:bwd", flops=10.0 * b * heads * t * t * d, bound="mfma", dtype="f32"):
            N.check(lib.smt_glow_attention_bwd(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(ek2), N.ptr(ev2), N.ptr(probs), N.ptr(_f(dctx)), N.ptr(dq),
                                               N.ptr(dk), N.ptr(dv), N.ptr(dek), N.ptr(dev), b, t, heads, d, window, drop.key,
                                               N.ptr(drop.key_dev), drop.thresh, drop.scale, N.ptr(ws), ws.numel(), N.stream_ptr()),
                    "smt_glow_attention_bwd")
        return dq, dk, dv, dek.view(ek_shape), dev.view(ev_shape), None, None, None, None


def rel_attention(q, k, v, emb_rel_k, emb_rel_v, lens, heads, window, drop=NO_DROP):
    """AttentionBlock.attention (submodules.py:463-512): self-attention with relative keys / values of one shared head
    (heads_share = True), padded positions filled with -1e4, dropout on the probabilities."""
    assert emb_rel_k.shape[0] == 1, "heads_share = True (the reference's default)"
    return _RelAttention.apply(q, k, v, emb_rel_k, emb_rel_v, _lens(lens), heads, window, drop)


@torch.no_grad()
def prior_logp(x_m, x_logs, z):
    """logp [B, Tx, Ty] of glow_tts.py:87-95 (x_logs None = zeros)."""
    x_m, z = _f(x_m), _f(z)
    b, tx, d = x_m.shape
    ty = z.shape[1]
    logp = torch.empty(b, tx, ty, device=z.device)
    with profiler.region("glow_prior_logp", flops=4.0 * b * tx * ty * d, bound="mfma", dtype="f32"):
        N.check(N.lib().smt_glow_prior_logp(N.ptr(x_m), N.ptr(None if x_logs is None else _f(x_logs)), N.ptr(z), N.ptr(logp), b, tx, ty, d,
                                            N.stream_ptr()), "smt_glow_prior_logp")
    return logp


@torch.no_grad()
def align_index(path):
    path = _f(path)
    b, tx, ty = path.shape
    idx = torch.empty(b, ty, dtype=torch.int32, device=path.device)
    dur = torch.empty(b, tx, device=path.device)
    N.check(N.lib().smt_glow_align_index(N.ptr(path), N.ptr(idx), N.ptr(dur), b, tx, ty, N.stream_ptr()), "smt_glow_align_index")
    return idx, dur


class _AlignGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = _f(x)
        b, tx, d = x.shape
        ty = idx.shape[1]
        z = torch.empty(b, ty, d, device=x.device)
        N.check(N.lib().smt_glow_align_gather(N.ptr(x), N.ptr(idx), N.ptr(z), b, tx, ty, d, N.stream_ptr()), "smt_glow_align_gather")
        ctx.save_for_backward(idx)
        ctx.tx = tx
        return z

    @staticmethod
    def backward(ctx, dz):
        (idx,) = ctx.saved_tensors
        dz = _f(dz)
        b, ty, d = dz.shape
        dx = torch.empty(b, ctx.tx, d, device=dz.device)
        N.check(N.lib().smt_glow_align_scatter(N.ptr(dz), N.ptr(idx), N.ptr(dx), b, ctx.tx, ty, d, N.stream_ptr()), "smt_glow_align_scatter")
        return dx, None


def align_gather(x, idx):
    """z[b, j] = x[b, idx[b, j]] -- ``torch.matmul(x_m, attn)`` of glow_tts.py:100-101 for a 0/1 monotonic path."""
    return _AlignGather.apply(x, idx)


class _MleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, z_m, z_logs, logdet_sum, denom):
        z, z_m = _f(z), _f(z_m)
        zl = None if z_logs is None else _f(z_logs)
        lib = N.lib()
        sums = torch.empty(2, device=z.device)
        ws = _ws(lib.smt_glow_mle_workspace_bytes(z.numel()), z.device)
        N.check(lib.smt_glow_mle_sums(N.ptr(z), N.ptr(z_m), N.ptr(zl), z.numel(), N.ptr(sums), N.ptr(ws), ws.numel(), N.stream_ptr()),
                "smt_glow_mle_sums")
        ctx.save_for_backward(z, z_m, zl if zl is not None else torch.empty(0), denom)
        ctx.has_logs = zl is not None
        return 0.5 * math.log(2 * math.pi) + (sums[0] + 0.5 * sums[1] - logdet_sum) / denom

    @staticmethod
    def backward(ctx, g):
        z, z_m, zl, denom = ctx.saved_tensors
        zl = zl if ctx.has_logs else None
        coef = (g / denom).reshape(1).float().contiguous()
        dz, dzm = torch.empty_like(z), torch.empty_like(z_m)
        dzl = torch.empty_like(z) if zl is not None else None
        N.check(N.lib().smt_glow_mle_bwd(N.ptr(z), N.ptr(z_m), N.ptr(zl), N.ptr(coef), z.numel(), N.ptr(dz), N.ptr(dzm), N.ptr(dzl), N.stream_ptr()),
                "smt_glow_mle_bwd")
        return dz, dzm, dzl, -g / denom, None


def mle_loss(z, z_m, z_logs, logdet_sum, denom):
    """0.5 log 2 pi + (sum z_logs + 0.5 sum exp(-2 z_logs)(z - z_m)^2 - logdet) / denom  (glow_tts.py:115-119); denom = a
    device scalar (sum of the frame lengths times the channel count)."""
    return _MleLoss.apply(z, z_m, z_logs, logdet_sum, denom)


class _LengthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logw, dur, lens, denom):
        logw, dur = _f(logw), _f(dur)
        b, tx = logw.shape
        diff = torch.empty_like(logw)
        s = torch.empty(1, device=logw.device)
        N.check(N.lib().smt_glow_length_loss(N.ptr(logw), N.ptr(dur), N.ptr(lens), b, tx, N.ptr(diff), N.ptr(s), N.stream_ptr()),
                "smt_glow_length_loss")
        ctx.save_for_backward(diff, denom)
        return s[0] / denom

    @staticmethod
    def backward(ctx, g):
        diff, denom = ctx.saved_tensors
        return diff * (2.0 * g / denom), None, None, None


def length_loss(logw, durations, lens, denom):
    """sum_{t < len} (logw - log(1e-8 + durations))^2 / denom  (glow_tts.py:99, 120)."""
    return _LengthLoss.apply(logw, durations, _lens(lens), denom)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return _dropout_apply(x, drop)

    @staticmethod
    def backward(ctx, dy):
        return _dropout_apply(dy, ctx.drop), None


def _dropout_apply(x, drop):
    x = _f(x)
    y = torch.empty_like(x)
    N.check(N.lib().smt_glow_dropout(N.ptr(x), N.ptr(y), x.numel(), drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale, N.stream_ptr()),
            "smt_glow_dropout")
    return y


def dropout(x, drop=NO_DROP):
    """y[i] = x[i] * keep(i) / (1 - p) over the linear element index (nn.Dropout of DurationPredictor, submodules.py:629-633)."""
    return x if drop.thresh == 0 else _Dropout.apply(x, drop)
