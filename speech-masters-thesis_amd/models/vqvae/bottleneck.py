"""Vector-quantisation bottleneck with EMA codebook (reference models/vqvae/bottleneck.py).

Rows are channels-last already, so ``preprocess`` (bottleneck.py:92-116) is a view.
The nearest-code search, dequantisation, commit/fit terms, straight-through backward
and the codebook statistics / update all run in libsmt_hip.so (``smt_amd.vq``).  Under
data parallelism the per-code sums, counts and rank 0's revival rows travel in ONE
buffer through ONE all-reduce (the reference issues a broadcast and two all-reduces,
bottleneck.py:73-75; adding zeros is exact, so the result is identical).
"""
import math

import torch
import torch.distributed as dist
import torch.nn as nn

from smt_amd import vq


class BottleneckBlock(nn.Module):
    def __init__(self, k_bins: int, emb_width: int, mu: float, threshold: float):
        super().__init__()
        self.k_bins, self.emb_width, self.mu, self.threshold = k_bins, emb_width, mu, threshold
        self.reset_k()

    def reset_k(self):
        self.init = False
        self.k_sum = None
        self.k_elem = None
        self.register_buffer("k", torch.zeros(self.k_bins, self.emb_width))
        self._prep, self._prep_key = None, None

    # -- derived data of the nearest-code search (centred bf16 split of the codes, norms): rebuilt only when `k`
    #    changes.  update_k refreshes it inside smt_vq_ema_apply; every other writer of `k` invalidates it. ----------
    def _invalidate_prep(self):
        self._prep_key = None

    def _search_prep(self):
        key = (self.k.data_ptr(), self.k._version)
        if self._prep_key != key:
            self._prep = vq.prepare(self.k, self._prep)
            self._prep_key = key
        return self._prep

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._invalidate_prep()

    # -- random rows for init / dead-code revival (bottleneck.py:26-33, :40, :69-70) -----------
    def _random_rows(self, rows, row_mask):
        """K rows drawn without replacement from the unmasked rows, without a host sync:
        sort uniform keys (masked rows pushed to +inf) and take the first K.  With fewer
        than K valid rows the selection wraps around and N(0, 0.01/sqrt(D)) jitter is
        added, which is what tiling does in the reference."""
        n = rows.shape[0]
        keys = torch.rand(n, device=rows.device)
        if row_mask is not None:
            keys = torch.where(row_mask != 0, keys, torch.full_like(keys, float("inf")))
            n_valid = (row_mask != 0).sum().clamp(min=1)
        else:
            n_valid = torch.tensor(n, device=rows.device)
        order = torch.argsort(keys)
        pick = order[torch.arange(self.k_bins, device=rows.device) % n_valid]
        out = rows[pick]
        jitter = (n_valid < self.k_bins).to(rows.dtype) * (0.01 / math.sqrt(self.emb_width))
        return out + jitter * torch.randn_like(out)

    def init_k(self, rows, row_mask=None, k_rand=None):
        self.init = True
        if k_rand is None:
            k_rand = self._random_rows(rows, row_mask)
        if dist.is_initialized():
            dist.broadcast(k_rand, 0)
        self.k = k_rand.clone()
        self._invalidate_prep()
        self.k_sum = self.k.clone()
        self.k_elem = torch.ones(self.k_bins, device=self.k.device)

    def restore_k(self, num_tokens=None, threshold=1.0):
        self.init = True
        self._invalidate_prep()
        self.k_sum = self.k.clone()
        self.k_elem = torch.ones(self.k_bins, device=self.k.device)
        if num_tokens is not None:
            expected = num_tokens / self.k_bins
            self.k_elem.mul_(expected)
            self.k_sum.mul_(expected)
        self.threshold = threshold

    @torch.no_grad()
    def update_k(self, rows, idx, row_mask, k_rand=None):
        kb, d = self.k_bins, self.emb_width
        stats = torch.empty(vq.ema_stats_numel(kb, d), device=rows.device, dtype=torch.float32)
        vq.ema_accumulate(rows, idx, row_mask, kb, stats)
        revival = stats[kb * d + kb:].view(kb, d)
        if (not dist.is_initialized()) or dist.get_rank() == 0:
            revival.copy_(self._random_rows(rows, row_mask) if k_rand is None else k_rand)
        else:
            revival.zero_()
        if dist.is_initialized():
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        m, self._prep = vq.ema_apply(self.k, self.k_sum, self.k_elem, stats, revival, self.mu, self.threshold, self._prep)
        self._prep_key = (self.k.data_ptr(), self.k._version)     # written by pointer: the version did not move
        return dict(entropy=m[0], used_curr=m[1], usage=m[2], dk=m[3])

    @torch.no_grad()
    def encode(self, x, lens):
        """x [B, T, D] -> codes [B, T] (bottleneck.py:147-158)."""
        b, t, d = x.shape
        idx, _, _, _ = vq.vq_forward_raw(x.reshape(b * t, d).float().contiguous(), self.k, None, want_xd=False,
                                         prep=self._search_prep())
        return idx.view(b, t)

    def decode(self, codes):
        """codes [B, T] -> [B, T, D] (bottleneck.py:160-169)."""
        return torch.nn.functional.embedding(codes, self.k)

    def forward(self, x, lens, update_k=True, k_rand=None, k_rand_init=None):
        b, t, d = x.shape
        rows = x.reshape(b * t, d).float().contiguous()
        steps = torch.arange(t, device=x.device)
        row_mask = (steps[None, :] < lens[:, None]).to(torch.float32).reshape(b * t)
        if update_k and not self.init:
            self.init_k(rows.detach(), row_mask, k_rand_init)
        # update_k rewrites self.k in place afterwards; backward reads the quantised rows from x_d, not from k
        x_d, idx, commit, fit = vq.vq_straight_through(rows, self.k, row_mask, detach_quantised=False,
                                                       prep=self._search_prep())
        metrics = dict(fit=fit)
        if update_k:
            metrics.update(self.update_k(rows.detach(), idx, row_mask, k_rand))
        return idx.view(b, t), x_d.view(b, t, d), commit, metrics


class Bottleneck(nn.Module):
    """Per-level wrapper (bottleneck.py:204-238); one level survives the VQVAE hack."""

    def __init__(self, l_bins, emb_width, mu, levels, threshold):
        super().__init__()
        self.levels = levels
        self.level_blocks = nn.ModuleList(BottleneckBlock(l_bins, emb_width, mu, threshold) for _ in range(levels))

    def encode(self, xs, lens):
        return [blk.encode(x, ln) for blk, x, ln in zip(self.level_blocks, xs, lens)]

    def decode(self, zs, start_level=0, end_level=None):
        end_level = self.levels if end_level is None else end_level
        return [blk.decode(z) for blk, z in zip(self.level_blocks[start_level:end_level], zs)]

    def forward(self, xs, lens, **vq_kwargs):
        zs, xqs, commits, metrics = [], [], [], []
        for level in range(self.levels):
            z, xq, commit, metric = self.level_blocks[level](xs[level], lens[level], update_k=self.training,
                                                             **vq_kwargs)
            if not self.training:
                xq = xq.detach()  # eval: no straight-through path into the encoder (bottleneck.py:230-233)
            zs.append(z); xqs.append(xq); commits.append(commit)
            if self.training:
                metrics.append(metric)
        return zs, xqs, commits, metrics
