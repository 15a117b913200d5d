"""Causal TransformerLM over VQ codes (reference models/transformer_lm/transformer_lm.py:32-155).

Same constructor keys, parameter names (a reference state_dict loads as is), ``forward`` / ``sample`` /
``reconstruct`` / ``load_vqvae`` surface and loss definition as the reference, which builds the stack from
``torch.nn.TransformerEncoder``.  Here a layer is five library GEMMs with everything between them in the HIP kernels
of csrc/lm.hip (attention core, add + dropout + LayerNorm, bias + ReLU + dropout, embedding, cross entropy); activations
are batch-major [B, L, d] instead of the reference's [L, B, d].  Dropout uses the counter-based generator keyed by
(step counter, site): site 0 = positional-encoding dropout, 1 + 4 i + {0, 1, 2, 3} = layer i's attention-weight,
attention-output, feed-forward-inner and feed-forward-output dropouts.
"""
import copy
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from models.base import TokenToWaveformModel
from smt_amd import lm as K


class PositionalEncoding(nn.Module):
    """Sinusoidal table ``pe`` [max_len, 1, d_model] (transformer_lm.py:14-29); applied inside the embedding kernel."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.p = dropout
        position = torch.arange(max_len, dtype=torch.float32)[:, None]
        div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def table(self):
        return self.pe.view(self.pe.shape[0], self.pe.shape[2])


class _SelfAttention(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names and initialisation."""

    def __init__(self, d_model, nhead):
        super().__init__()
        assert d_model % nhead == 0
        self.num_heads = nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayer(nn.Module):
    """Post-norm encoder layer with nn.TransformerEncoderLayer's parameter names."""

    def __init__(self, d_model, nhead, dim_feedforward, dropout, layer_norm_eps=1e-5):
        super().__init__()
        self.self_attn = _SelfAttention(d_model, nhead)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.p = dropout

    def forward(self, x, lens, causal, seed, site0, kd=None):
        sa, tr, p = self.self_attn, self.training, self.p
        qkv = F.linear(x, sa.in_proj_weight, sa.in_proj_bias)
        ctx = K.attention(qkv, lens, sa.num_heads, causal, K.Drop(p, tr, seed, site0, kd))
        a = F.linear(ctx, sa.out_proj.weight)                      # its bias is added (and differentiated) inside add_layer_norm
        x = K.add_layer_norm(x, a, self.norm1.weight, self.norm1.bias, self.norm1.eps, K.Drop(p, tr, seed, site0 + 1, kd),
                             h_bias=sa.out_proj.bias)
        f = K.bias_relu_dropout_(F.linear(x, self.linear1.weight), self.linear1.bias, K.Drop(p, tr, seed, site0 + 2, kd))
        f = F.linear(f, self.linear2.weight)
        return K.add_layer_norm(x, f, self.norm2.weight, self.norm2.bias, self.norm2.eps, K.Drop(p, tr, seed, site0 + 3, kd),
                                h_bias=self.linear2.bias)


class _Encoder(nn.Module):
    """``layers`` + final ``norm`` as in nn.TransformerEncoder, which deep-copies ONE initialised layer: every layer
    starts from the same weights, and so do these."""

    def __init__(self, layer, num_layers, norm):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.norm = norm


class TransformerLM(TokenToWaveformModel):

    PAD = 0     # <pad> token
    BOS = 1     # <bos> token
    OFFSET = 2  # number of special tokens the original vocabulary is shifted by

    def __init__(self, config):
        super().__init__()
        m = config.model
        assert m.embed_dim == m.d_model, "the embedding feeds the encoder directly"
        if m.get("norm_first", False) or m.get("activation", "relu") != "relu":
            raise ValueError("native TransformerLM path: post-norm layers with ReLU (the reference's construction)")
        self.d_model = m.d_model
        self.embedding = nn.Embedding(m.vocab_size + TransformerLM.OFFSET, m.embed_dim, padding_idx=TransformerLM.PAD)
        self.pos_encoding = PositionalEncoding(m.d_model, m.dropout, m.max_len)
        # the reference's layers keep nn.TransformerEncoderLayer's default eps; only the final norm takes the config's
        layer = _EncoderLayer(m.d_model, m.nhead, m.dim_feedforward, m.dropout)
        self.transformer = _Encoder(layer, m.num_layers, nn.LayerNorm(m.d_model, eps=float(m.layer_norm_eps)))
        self.classifier = nn.Linear(m.d_model, m.vocab_size)
        self.vqvae = TransformerLM.load_vqvae(m.vqvae.log_dir, m.vqvae.ckpt_num)
        self.loss_type = m.loss_type
        if m.loss_type == "ce":
            self.loss = None                                    # native kernel (smt_lm_ce_fwd / _bwd)
        elif m.loss_type == "mmi":
            from models.transformer_lm.losses import MaximumMutualInformationLoss
            self.loss = MaximumMutualInformationLoss(num_classes=m.vocab_size)
        elif m.loss_type == "focal":
            from models.transformer_lm.losses import FocalLoss
            self.loss = FocalLoss(gamma=10.0, reduction="mean")
        else:
            raise ValueError(f"Loss function {m.loss_type} not supported")
        self._drop_seed = 0
        self._seed_dev = self._keys_dev = None          # device-resident dropout keys (enable_device_keys)

    def enable_device_keys(self, flag=True):
        """Keep the dropout step counter and the per-site keys in device memory (smt_lm_make_keys) instead of passing keys
        by value: a captured hipGraph of the train step then draws fresh masks on every replay (smt_amd/graph.py).  The
        masks are the same either way -- the host counter `_drop_seed` and the device one advance together."""
        if not flag:
            self._seed_dev = self._keys_dev = None
            return
        dev = self.embedding.weight.device
        self._seed_dev = torch.tensor([self._drop_seed & 0x7FFFFFFF], dtype=torch.int32, device=dev)
        self._keys_dev = torch.zeros(1 + 4 * len(self.transformer.layers), dtype=torch.int32, device=dev)
        K.make_keys(self._seed_dev, self._keys_dev)

    @staticmethod
    def load_vqvae(log_dir, ckpt_num):
        """Frozen-architecture VQ-VAE pieces for audio reconstruction (transformer_lm.py:84-98): the run's config.yaml
        and ckpts/ckpt.<n>.pt -> {"bottleneck": level block, "decoder": decoder} (they stay trainable parameters of
        this model, as in the reference)."""
        from models.vqvae.vqvae import VQVAE
        from utils import config as cfglib
        config = cfglib.load(os.path.join(log_dir, "config.yaml"))
        ckpt = torch.load(os.path.join(log_dir, "ckpts", f"ckpt.{ckpt_num}.pt"), map_location="cpu", weights_only=True)
        vqvae = VQVAE(config)
        vqvae.load_state_dict(ckpt["model"])
        block = vqvae.bottleneck.level_blocks[vqvae.LEVEL]
        holder = nn.ModuleDict({"bottleneck": block, "decoder": vqvae.decoders[vqvae.LEVEL]})
        holder.compute_dtype = vqvae.compute_dtype
        return holder

    def reconstruct(self, q, mask):
        """codes q [B, T'] (no special-token offset), mask [B, 1, T'] (a length prefix) -> audio [B, T]."""
        lens = mask.reshape(mask.shape[0], -1).to(torch.int32).sum(-1).to(torch.int32)
        with torch.no_grad():
            xq = self.vqvae["bottleneck"].decode(q)
            keep = (torch.arange(q.shape[1], device=q.device)[None, :] < lens[:, None]).unsqueeze(-1)
            y, y_lens = self.vqvae["decoder"]((xq * keep).to(self.vqvae.compute_dtype), lens)
            keep_t = torch.arange(y.shape[1], device=y.device)[None, :] < y_lens[:, None]
        return (y * keep_t).float()

    def logits(self, x, lens, causal=True):
        """tokens [B, L] int64 (+ int32 lengths or None) -> next-token logits [B, L, vocab]."""
        seed, tr, kd = self._drop_seed, self.training, self._keys_dev
        h = K.embed(x, self.embedding.weight, self.pos_encoding.table(), K.Drop(self.pos_encoding.p, tr, seed, 0, kd),
                    TransformerLM.PAD)
        for i, layer in enumerate(self.transformer.layers):
            h = layer(h, lens, causal, seed, 1 + 4 * i, kd)
        norm = self.transformer.norm
        h = K.add_layer_norm(h, None, norm.weight, norm.bias, norm.eps)
        return F.linear(h, self.classifier.weight, self.classifier.bias)

    def forward(self, x, x_lengths, y, y_lengths, speaker=None):
        b, l = x.shape
        lens = x_lengths.to(device=x.device, dtype=torch.int32)
        if self.training:
            self._drop_seed += 1                                # one fresh set of dropout masks per training step
            if self._keys_dev is not None:                      # the same counter on the device, and the keys derived from it
                self._seed_dev.add_(1)
                K.make_keys(self._seed_dev, self._keys_dev)
        xh = self.logits(x, lens, causal=True)
        # next-token targets (transformer_lm.py:121-126): position t predicts x[t + 1]; pads / specials are not scored
        nxt = x[:, 1:]
        target = torch.full_like(x, -1)
        target[:, :-1] = torch.where(nxt >= TransformerLM.OFFSET, nxt - TransformerLM.OFFSET, -1)
        if self.loss is None:
            loss, accuracy, _ = K.cross_entropy(xh, target)
        else:
            rows = target.reshape(-1) >= 0
            scored, tgt = xh.reshape(b * l, -1)[rows], target.reshape(-1)[rows]
            loss = self.loss(scored, tgt)
            accuracy = (scored.argmax(1) == tgt).sum().float() / rows.sum()
        if not self.training:
            keep = torch.arange(l - 1, device=x.device)[None, :] < lens[:, None]
            yh = self.reconstruct(xh[:, :-1, :].argmax(-1), keep[:, None, :])
        else:
            yh = None
        return {"loss": loss, "yh": yh}, {"accuracy": accuracy}

    @torch.no_grad()
    def sample(self, batch_size, n_steps, device="cuda", sigma=1.0):
        """Ancestral sampling (transformer_lm.py:137-155).  As in the reference every step re-runs the whole prefix WITHOUT
        the causal mask (mask=None there), so a key/value cache cannot reproduce it; the step is one pass of `logits`."""
        assert sigma > 0, "Temperature scalar must be positive"
        q = torch.full((batch_size, 1), TransformerLM.BOS, dtype=torch.long, device=device)
        for _ in range(n_steps):
            probs = F.softmax(self.logits(q, None, causal=False)[:, -1, :] / sigma, dim=-1)
            q = torch.cat([q, torch.multinomial(probs, 1)], dim=-1)
        q = q[:, 1:]
        return self.reconstruct(q, torch.ones_like(q).unsqueeze(1)), q
