"""Autograd bindings of the TransformerLM kernels (csrc/lm.hip, include/smt_hip.h "TransformerLM").

Everything between the dense projections of the reference's nn.TransformerEncoder stack
(models/transformer_lm/transformer_lm.py:54-66) runs in these kernels; the projections themselves are library GEMMs
(torch.nn.functional.linear -> hipBLASLt).  Activations are [batch, len, dim] fp32; dropout masks come from the
counter-based generator (smt_hip.h "dropout") keyed per (step seed, site), so backward recomputes them instead of
storing a mask per site.
"""
import math

import torch

from . import native as N
from . import profiler
from .convops import dropout_key


class Drop:
    """(key, thresh16, scale) of one dropout site; p = 0 or eval -> identity.  ``keys_dev`` (a device uint32/int32 tensor
    of per-site keys written by ``make_keys``) makes the kernels read the key from device memory instead of taking it by
    value: what a captured graph needs, since replays cannot change by-value arguments."""
    __slots__ = ("key", "thresh", "scale", "key_dev")

    def __init__(self, p=0.0, training=False, seed=0, site=0, keys_dev=None):
        self.key_dev = None
        if training and p > 0.0:
            self.key, self.thresh, self.scale = dropout_key(seed, site), int(round(p * 65536.0)), 1.0 / (1.0 - p)
            if keys_dev is not None:
                self.key_dev = keys_dev[site:site + 1]
        else:
            self.key, self.thresh, self.scale = 0, 0, 1.0


def make_keys(seed_dev, keys_dev):
    """keys_dev[s] = dropout_key(seed_dev[0], s) computed on the device (smt_lm_make_keys)."""
    assert seed_dev.is_cuda and keys_dev.is_cuda and seed_dev.dtype == torch.int32 and keys_dev.dtype == torch.int32
    N.check(N.lib().smt_lm_make_keys(N.ptr(seed_dev), N.ptr(keys_dev), keys_dev.numel(), N.stream_ptr()), "smt_lm_make_keys")


NO_DROP = Drop()


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda, "TransformerLM kernels take fp32 device tensors"
    return t.contiguous()


class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, weight, pe, mul, drop, padding_idx):
        b, l = tokens.shape
        d = weight.shape[1]
        tokens = tokens.contiguous()
        out = torch.empty(b, l, d, device=weight.device, dtype=torch.float32)
        N.check(N.lib().smt_lm_embed_fwd(N.ptr(tokens), N.ptr(_f32(weight)), N.ptr(_f32(pe)), N.ptr(out), b, l, d, mul, drop.key, N.ptr(drop.key_dev),
                                         drop.thresh, drop.scale, N.stream_ptr()), "smt_lm_embed_fwd")
        ctx.save_for_backward(tokens)
        ctx.meta = (weight.shape[0], d, mul, drop, padding_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tokens,) = ctx.saved_tensors
        rows, d, mul, drop, padding_idx = ctx.meta
        b, l = tokens.shape
        demb = torch.empty(rows, d, device=dout.device, dtype=torch.float32)
        N.check(N.lib().smt_lm_embed_bwd(N.ptr(tokens), N.ptr(_f32(dout)), N.ptr(demb), b, l, d, rows, mul, drop.key, N.ptr(drop.key_dev), drop.thresh,
                                         drop.scale, padding_idx, N.stream_ptr()), "smt_lm_embed_bwd")
        return None, demb, None, None, None, None


def embed(tokens, weight, pe, drop=NO_DROP, padding_idx=0):
    """dropout(weight[tokens] * sqrt(dim) + pe[:len])  (transformer_lm.py:114-116, :27-29); tokens [batch, len] int64."""
    assert tokens.dtype == torch.int64 and pe.shape[0] >= tokens.shape[1] and pe.shape[-1] == weight.shape[1]
    return _Embed.apply(tokens, weight, pe, math.sqrt(weight.shape[1]), drop, padding_idx)


def _attn_flops(b, heads, l, causal, matmuls):
    """Algorithmic FLOPs of `matmuls` 32-channel products over the visible (query, key) pairs (full lengths): forward 2
    (scores, context), backward 5 (scores, dP, dq, dk, dv; the two backward kernels each recompute scores and dP, which
    is not counted)."""
    pairs = l * (l + 1) // 2 if causal else l * l
    return b * heads * pairs * 64 * matmuls


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, lens, heads, causal, drop):
        b, l, d3 = qkv.shape
        d = d3 // 3
        assert d == heads * 32, "the attention kernel is built for head dim 32"
        qkv = _f32(qkv)
        out = torch.empty(b, l, d, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(b, heads, l, device=qkv.device, dtype=torch.float32)
        with profiler.region("lm_attention:fwd", flops=_attn_flops(b, heads, l, causal, 2), bound="mfma", dtype="f32"):
            N.check(N.lib().smt_lm_attention_fwd(N.ptr(qkv), N.ptr(lens), N.ptr(out), N.ptr(lse), b, l, heads, int(causal),
                                                 drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale, N.stream_ptr()), "smt_lm_attention_fwd")
        ctx.save_for_backward(qkv, lens, out, lse)
        ctx.meta = (heads, int(causal), drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, lens, out, lse = ctx.saved_tensors
        heads, causal, drop = ctx.meta
        b, l, _ = qkv.shape
        dqkv = torch.empty_like(qkv)
        delta = torch.empty_like(lse)
        dout = _f32(dout)
        with profiler.region("lm_attention:bwd", flops=_attn_flops(b, heads, l, causal, 5), bound="mfma", dtype="f32"):
            N.check(N.lib().smt_lm_attention_bwd(N.ptr(qkv), N.ptr(lens), N.ptr(out), N.ptr(lse), N.ptr(dout), N.ptr(dqkv),
                                                 N.ptr(delta), b, l, heads, causal, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale,
                                                 N.stream_ptr()), "smt_lm_attention_bwd")
        return dqkv, None, None, None, None


def attention(qkv, lens, heads, causal=True, drop=NO_DROP):
    """Self-attention core: qkv [batch, len, 3 dim] -> [batch, len, dim]; lens [batch] int32 (keys >= lens[b] masked) or
    None; causal adds the triu(-inf, 1) mask of the reference's forward (:111)."""
    if lens is not None:
        assert lens.dtype == torch.int32 and lens.is_cuda
    return _Attention.apply(qkv, lens, heads, causal, drop)


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, h_bias, gamma, beta, eps, drop):
        src = x if x is not None else h
        d = src.shape[-1]
        rows = src.numel() // d
        x = _f32(x) if x is not None else None
        h = _f32(h) if h is not None else None
        y = torch.empty_like(src, memory_format=torch.contiguous_format)
        stats = torch.empty(rows, 2, device=src.device, dtype=torch.float32)
        n_in = (x is not None) + (h is not None)
        with profiler.region("lm_add_ln:fwd", nbytes=(n_in + 1) * rows * d * 4, bound="hbm"):
            N.check(N.lib().smt_lm_add_ln_fwd(N.ptr(x), N.ptr(h), N.ptr(h_bias), N.ptr(_f32(gamma)), N.ptr(_f32(beta)), N.ptr(y),
                                              N.ptr(stats), rows, d, eps, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale, N.stream_ptr()),
                    "smt_lm_add_ln_fwd")
        ctx.save_for_backward(x, h, h_bias, gamma, stats)
        ctx.meta = (rows, d, drop)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, h_bias, gamma, stats = ctx.saved_tensors
        rows, d, drop = ctx.meta
        lib = N.lib()
        dy = _f32(dy)
        dx = torch.empty_like(dy) if x is not None and ctx.needs_input_grad[0] else None
        dh = torch.empty_like(dy) if h is not None and ctx.needs_input_grad[1] else None
        dparams = torch.empty(3, d, device=dy.device, dtype=torch.float32)
        ws_bytes = lib.smt_lm_add_ln_bwd_workspace_bytes(rows, d)
        ws = torch.empty(ws_bytes, device=dy.device, dtype=torch.uint8)
        n_io = 1 + (x is not None) + (h is not None) + (dx is not None) + (dh is not None)
        with profiler.region("lm_add_ln:bwd", nbytes=n_io * rows * d * 4, bound="hbm"):
            N.check(lib.smt_lm_add_ln_bwd(N.ptr(x), N.ptr(h), N.ptr(h_bias), N.ptr(dy), N.ptr(_f32(gamma)), N.ptr(stats), N.ptr(dx),
                                          N.ptr(dh), N.ptr(dparams), rows, d, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale, N.ptr(ws),
                                          ws_bytes, N.stream_ptr()), "smt_lm_add_ln_bwd")
        return dx, dh, (dparams[2] if h_bias is not None else None), dparams[0], dparams[1], None, None


def add_layer_norm(x, h, gamma, beta, eps=1e-5, drop=NO_DROP, h_bias=None):
    """LayerNorm(x + dropout(h + h_bias)) (post-norm sub-layer of nn.TransformerEncoderLayer; h_bias = the bias of the
    projection that produced h, whose gradient then comes out of this op's backward); h = None -> LayerNorm(x)."""
    assert h_bias is None or (h is not None and h_bias.dtype == torch.float32 and h_bias.is_contiguous())
    return _AddLayerNorm.apply(x, h, h_bias, gamma, beta, eps, drop)


class _BiasReluDrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, bias, drop):
        d = h.shape[-1]
        rows = h.numel() // d
        h = _f32(h)
        with profiler.region("lm_bias_relu:fwd", nbytes=2 * rows * d * 4, bound="hbm"):
            N.check(N.lib().smt_lm_bias_relu_fwd(N.ptr(h), N.ptr(_f32(bias)), rows, d, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale,
                                                 N.stream_ptr()), "smt_lm_bias_relu_fwd")
        ctx.mark_dirty(h)
        ctx.save_for_backward(h)
        ctx.meta = (rows, d, drop)
        return h

    @staticmethod
    def backward(ctx, da):
        (a,) = ctx.saved_tensors
        rows, d, drop = ctx.meta
        lib = N.lib()
        da = _f32(da)
        dh = torch.empty_like(da)
        dbias = torch.empty(d, device=a.device, dtype=torch.float32)
        ws_bytes = lib.smt_lm_bias_relu_bwd_workspace_bytes(rows, d)
        ws = torch.empty(ws_bytes, device=a.device, dtype=torch.uint8)
        with profiler.region("lm_bias_relu:bwd", nbytes=3 * rows * d * 4, bound="hbm"):
            N.check(lib.smt_lm_bias_relu_bwd(N.ptr(a), N.ptr(da), N.ptr(dh), N.ptr(dbias), rows, d, drop.key, N.ptr(drop.key_dev), drop.thresh, drop.scale,
                                             N.ptr(ws), ws_bytes, N.stream_ptr()), "smt_lm_bias_relu_bwd")
        return dh, dbias, None


def bias_relu_dropout_(h, bias, drop=NO_DROP):
    """In place: dropout(relu(h + bias)) on the (bias-free) output of linear1 (feed-forward of the encoder layer)."""
    return _BiasReluDrop.apply(h, bias, drop)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        v = logits.shape[-1]
        rows = logits.numel() // v
        logits = _f32(logits)
        target = target.contiguous()
        row_out = torch.empty(rows, 2, device=logits.device, dtype=torch.float32)
        lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
        N.check(N.lib().smt_lm_ce_fwd(N.ptr(logits), N.ptr(target), N.ptr(row_out), N.ptr(lse), rows, v, N.stream_ptr()),
                "smt_lm_ce_fwd")
        count = (target >= 0).sum().to(torch.float32)
        sums = row_out.sum(0)
        ctx.save_for_backward(logits, target, lse, count)
        ctx.mark_non_differentiable(count)
        loss, acc = sums[0] / count, sums[1] / count
        ctx.mark_non_differentiable(acc)
        return loss, acc, count

    @staticmethod
    def backward(ctx, dloss, _dacc, _dcount):
        logits, target, lse, count = ctx.saved_tensors
        v = logits.shape[-1]
        rows = logits.numel() // v
        coef = (dloss.to(torch.float32) / count).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        N.check(N.lib().smt_lm_ce_bwd(N.ptr(logits), N.ptr(target), N.ptr(lse), N.ptr(coef), N.ptr(dlogits), rows, v,
                                      N.stream_ptr()), "smt_lm_ce_bwd")
        return dlogits, None


def cross_entropy(logits, target):
    """(mean CE over rows with target >= 0, accuracy over the same rows, their number) -- transformer_lm.py:121-128 with
    the reference's boolean row selection turned into target = -1 on the rows it drops."""
    assert target.dtype == torch.int64
    return _CrossEntropy.apply(logits, target)
