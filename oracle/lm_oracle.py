"""TEST INFRASTRUCTURE -- CPU restatement of the reference's TransformerLM forward (models/transformer_lm/
transformer_lm.py:109-135) with torch.nn.TransformerEncoder's layer written out operation by operation
(post-norm nn.TransformerEncoderLayer: self-attention with the additive causal mask and the key-padding mask, dropout on
the attention weights and on both sub-layer outputs, ReLU feed-forward with inner dropout, final LayerNorm), so that the
product's counter-based dropout masks (include/smt_hip.h "dropout") can be injected at the exact sites.  Plain torch on
the CPU, any float dtype (tests use float64 as the yardstick); gradients come from torch autograd over this forward.

Pinned by tests/golden/transformer_lm.npz: logits, loss, accuracy and every parameter gradient captured from the
reference's own class (dropout 0 -- its dropout draws from torch's global RNG, which nothing else reproduces).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle.vqvae_oracle import dropout_keep_ntc

PAD, BOS, OFFSET = 0, 1, 2      # transformer_lm.py:34-36


def positional_table(max_len: int, d_model: int) -> torch.Tensor:
    """PositionalEncoding.pe without its singleton batch axis (transformer_lm.py:20-25)."""
    position = torch.arange(max_len, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe


class CounterDropout:
    """The product's dropout sites: mask of element i of the dropped tensor (row-major over its own shape) under key
    (seed, site); site 0 = embedding, 1 + 4 layer + {0 attention weights [B, H, L, L], 1 attention output, 2 feed-forward
    inner, 3 feed-forward output}.  p = 0 -> identity."""

    def __init__(self, seed: int = 0, p: float = 0.0):
        self.seed, self.p = seed, p

    def __call__(self, site: int, x: torch.Tensor) -> torch.Tensor:
        if self.p <= 0.0:
            return x
        keep = dropout_keep_ntc(self.seed, site, 1, 1, x.numel(), self.p).reshape(tuple(x.shape))
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(self.p))
        return x * torch.from_numpy(keep.astype(np.float32) * scale).to(x.dtype)


def attention_core(qkv: torch.Tensor, lens: Optional[torch.Tensor], heads: int, causal: bool, drop, site: int) -> torch.Tensor:
    """nn.MultiheadAttention between in_proj and out_proj: qkv [B, L, 3 d] -> [B, L, d]."""
    b, l, d3 = qkv.shape
    d = d3 // 3
    dh = d // heads
    q, k, v = (t.reshape(b, l, heads, dh).permute(0, 2, 1, 3) for t in qkv.split(d, dim=-1))
    s = (q / math.sqrt(dh)) @ k.transpose(-1, -2)                            # [B, H, L, L]
    if causal:
        s = s + torch.triu(torch.full((l, l), float("-inf"), dtype=s.dtype), diagonal=1)
    if lens is not None:
        pad = torch.arange(l)[None, :] >= lens[:, None]                      # ~sequence_mask(x_lengths) (:110,117)
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    w = drop(site, F.softmax(s, dim=-1))
    return (w @ v).permute(0, 2, 1, 3).reshape(b, l, d)


def encoder_layer(x, p: Dict[str, torch.Tensor], pre: str, lens, heads: int, causal: bool, drop, site0: int, eps: float = 1e-5):
    qkv = F.linear(x, p[pre + "self_attn.in_proj_weight"], p[pre + "self_attn.in_proj_bias"])
    a = attention_core(qkv, lens, heads, causal, drop, site0)
    a = F.linear(a, p[pre + "self_attn.out_proj.weight"], p[pre + "self_attn.out_proj.bias"])
    x = F.layer_norm(x + drop(site0 + 1, a), x.shape[-1:], p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
    f = drop(site0 + 2, F.relu(F.linear(x, p[pre + "linear1.weight"], p[pre + "linear1.bias"])))
    f = F.linear(f, p[pre + "linear2.weight"], p[pre + "linear2.bias"])
    return F.layer_norm(x + drop(site0 + 3, f), x.shape[-1:], p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)


def lm_logits(x: torch.Tensor, lens: Optional[torch.Tensor], p: Dict[str, torch.Tensor], heads: int, num_layers: int,
              final_eps: float = 1e-5, causal: bool = True, drop=None) -> torch.Tensor:
    """tokens [B, L] -> logits [B, L, vocab] (transformer_lm.py:113-119, batch-major)."""
    drop = drop or CounterDropout()
    emb = p["embedding.weight"]
    d = emb.shape[1]
    pe = positional_table(x.shape[1], d).to(emb.dtype)
    h = drop(0, F.embedding(x, emb) * math.sqrt(d) + pe[None])
    for i in range(num_layers):
        h = encoder_layer(h, p, f"transformer.layers.{i}.", lens, heads, causal, drop, 1 + 4 * i)
    h = F.layer_norm(h, (d,), p["transformer.norm.weight"], p["transformer.norm.bias"], final_eps)
    return F.linear(h, p["classifier.weight"], p["classifier.bias"])


def lm_loss(x: torch.Tensor, logits: torch.Tensor):
    """(mean cross entropy, accuracy) over the next-token positions whose target is a real code (:121-128)."""
    x_flat = x[:, 1:].flatten()
    xh_flat = logits[:, :-1, :].reshape(len(x_flat), -1)
    scored = x_flat >= OFFSET
    target = x_flat[scored] - OFFSET
    loss = F.cross_entropy(xh_flat[scored], target, reduction="mean")
    acc = (target == xh_flat[scored].argmax(1)).sum().to(logits.dtype) / scored.sum()
    return loss, acc


def init_params(vocab: int, d_model: int, heads: int, ff: int, num_layers: int, seed: int = 0, dtype=torch.float32):
    """Random parameters with the reference's names (not its initialisation: tests want every layer different)."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).to(dtype)

    p = {"embedding.weight": rnd(vocab + OFFSET, d_model, scale=1.0), "classifier.weight": rnd(vocab, d_model, scale=d_model ** -0.5),
         "classifier.bias": rnd(vocab, scale=0.1), "transformer.norm.weight": 1 + rnd(d_model, scale=0.1),
         "transformer.norm.bias": rnd(d_model, scale=0.1)}
    p["embedding.weight"][PAD] = 0
    for i in range(num_layers):
        pre = f"transformer.layers.{i}."
        p[pre + "self_attn.in_proj_weight"] = rnd(3 * d_model, d_model, scale=d_model ** -0.5)
        p[pre + "self_attn.in_proj_bias"] = rnd(3 * d_model, scale=0.1)
        p[pre + "self_attn.out_proj.weight"] = rnd(d_model, d_model, scale=d_model ** -0.5)
        p[pre + "self_attn.out_proj.bias"] = rnd(d_model, scale=0.1)
        p[pre + "linear1.weight"] = rnd(ff, d_model, scale=d_model ** -0.5)
        p[pre + "linear1.bias"] = rnd(ff, scale=0.1)
        p[pre + "linear2.weight"] = rnd(d_model, ff, scale=ff ** -0.5)
        p[pre + "linear2.bias"] = rnd(d_model, scale=0.1)
        for n in ("norm1", "norm2"):
            p[pre + n + ".weight"] = 1 + rnd(d_model, scale=0.1)
            p[pre + n + ".bias"] = rnd(d_model, scale=0.1)
    return p


def synthetic_tokens(batch: int, length: int, vocab: int, seed: int, ragged: bool = True):
    """Token batch as datasets/vqlatent.py collates it: <bos>, codes + OFFSET, <pad> up to the longest (int64), lengths."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.full((batch,), length, dtype=torch.int64)
    if ragged and batch > 1:
        lens[1:] = torch.randint(max(2, length // 2), length + 1, (batch - 1,), generator=g)
    x = torch.randint(OFFSET, vocab + OFFSET, (batch, length), generator=g)
    x[:, 0] = BOS
    x[torch.arange(length)[None, :] >= lens[:, None]] = PAD
    return x, lens
