"""Vector-quantiser ops on the HIP library (include/smt_hip.h, 'VQ' section).

Host-side counterpart of BottleneckBlock.quantize/dequantize/update_k
(reference models/vqvae/bottleneck.py:60-90, 126-145, 171-201).
"""
import torch

from . import native as N
from . import profiler


def prepare(codebook, prep=None):
    """Derived data of the nearest-code search for THIS codebook content (include/smt_hip.h, smt_vq_prepare): returns
    the persistent buffer to hand to vq_forward_raw / vq_straight_through / ema_apply."""
    assert codebook.dtype == torch.float32 and codebook.is_cuda
    k, d = codebook.shape
    lib = N.lib()
    nbytes = lib.smt_vq_prep_bytes(k, d)
    if prep is None or prep.numel() < nbytes or prep.device != codebook.device:
        prep = torch.empty(nbytes, dtype=torch.uint8, device=codebook.device)
    N.check(lib.smt_vq_prepare(N.ptr(codebook), k, d, N.ptr(prep), prep.numel(), N.stream_ptr()), "smt_vq_prepare")
    return prep


def vq_forward_raw(x, codebook, row_mask=None, want_xd=True, prep=None):
    """x [n, D] f32, codebook [K, D] f32, row_mask [n] f32|None ->
    (idx int64 [n], min_dist f32 [n], x_d f32 [n, D]|None, sums f32 [4]).  ``prep`` = prepare(codebook) of the same
    codebook content (None: rebuilt inside the call)."""
    assert x.dtype == torch.float32 and codebook.dtype == torch.float32
    n, d = x.shape
    k = codebook.shape[0]
    lib = N.lib()
    idx = torch.empty(n, dtype=torch.int64, device=x.device)
    min_dist = torch.empty(n, dtype=torch.float32, device=x.device)
    x_d = torch.empty_like(x) if want_xd else None
    sums = torch.empty(4, dtype=torch.float32, device=x.device)
    ws_bytes = lib.smt_vq_forward_workspace_bytes(n, k, d)
    ws = N.workspace.get(ws_bytes, x.device)
    # algorithmic bytes: 4D read + 8 idx + 4 min_dist (+ 4D x_d) per row, + codebook once
    nbytes = n * (4 * d + 12 + (4 * d if want_xd else 0)) + 4 * k * d
    # flops: the filter's three bf16 MFMA products per (row, code, dim) -- the work the kernel really issues
    with profiler.region("vq_forward", nbytes=nbytes, flops=3 * 2.0 * n * k * d, bound="hbm", dtype="bf16"):
        N.check(lib.smt_vq_forward(N.ptr(x), N.ptr(codebook), N.ptr(prep), N.ptr(row_mask), n, k, d, N.ptr(idx),
                                   N.ptr(min_dist), N.ptr(x_d), N.ptr(sums), N.ptr(ws), ws.numel(),
                                   N.stream_ptr()), "smt_vq_forward")
    return idx, min_dist, x_d, sums


class _VQStraightThrough(torch.autograd.Function):
    """(x, codebook, row_mask) -> (x_d*mask, idx, commit, fit); backward = straight-through
    + commit-loss gradient (bottleneck.py:194-201).  Backward reads the quantised rows from x_d, not from the
    codebook, so the caller may rewrite the codebook in place (update_k) before backward runs."""

    @staticmethod
    def forward(ctx, x, codebook, row_mask, detach_quantised, prep):
        x = x.contiguous()
        idx, min_dist, x_d, sums = vq_forward_raw(x, codebook, row_mask, prep=prep)
        n, d = x.shape
        k = codebook.shape[0]
        commit = sums[1] / (sums[2] * d)          # ||x_d - x||^2 over unmasked rows / (sum mask * D)
        fit = sums[0] / k                         # reference's [N]*[N,1] broadcast: sum over ALL rows / K
        ctx.save_for_backward(x, x_d, row_mask if row_mask is not None else torch.empty(0), sums)
        ctx.has_mask = row_mask is not None
        ctx.detach_quantised = detach_quantised
        ctx.mark_non_differentiable(idx, fit)
        return x_d, idx, commit, fit

    @staticmethod
    def backward(ctx, g_xd, g_idx, g_commit, g_fit):
        x, x_d, row_mask, sums = ctx.saved_tensors
        row_mask = row_mask if ctx.has_mask else None
        n, d = x.shape
        dy = None if (g_xd is None or ctx.detach_quantised) else g_xd.contiguous()
        gc = None if g_commit is None else g_commit.reshape(1).to(torch.float32).contiguous()
        dx = torch.empty_like(x)
        lib = N.lib()
        N.check(lib.smt_vq_backward(N.ptr(x), N.ptr(x_d), N.ptr(row_mask), N.ptr(dy), N.ptr(gc), N.ptr(sums), n, d,
                                    N.ptr(dx), N.stream_ptr()), "smt_vq_backward")
        return dx, None, None, None, None


def vq_straight_through(x, codebook, row_mask=None, detach_quantised=False, prep=None):
    return _VQStraightThrough.apply(x, codebook, row_mask, detach_quantised, prep)


def ema_stats_numel(k_bins, dim):
    """[K*D sums | K counts | K*D revival rows] -- one buffer, one all-reduce."""
    return k_bins * dim + k_bins + k_bins * dim


@torch.no_grad()
def ema_accumulate(x, idx, row_mask, k_bins, stats):
    n, d = x.shape
    lib = N.lib()
    ws_bytes = lib.smt_vq_ema_accumulate_workspace_bytes(n, k_bins, d)
    ws = N.workspace.get(ws_bytes, x.device)
    with profiler.region("vq_ema_accumulate", nbytes=n * (4 * d + 8) + 8 * k_bins * (d + 1), bound="hbm"):
        N.check(lib.smt_vq_ema_accumulate(N.ptr(x), N.ptr(idx), N.ptr(row_mask), n, k_bins, d, N.ptr(stats), N.ptr(ws),
                                          ws.numel(), N.stream_ptr()), "smt_vq_ema_accumulate")


@torch.no_grad()
def ema_apply(codebook, k_sum, k_elem, stats, k_rand, mu, threshold, prep=None):
    """In-place update of codebook / k_sum / k_elem; ``prep`` (see prepare) is refreshed for the new codebook in the
    same call.  Returns (metrics [4], prep)."""
    k, d = codebook.shape
    metrics = torch.empty(4, dtype=torch.float32, device=codebook.device)
    lib = N.lib()
    nbytes = lib.smt_vq_prep_bytes(k, d)
    if prep is None or prep.numel() < nbytes or prep.device != codebook.device:
        prep = torch.empty(nbytes, dtype=torch.uint8, device=codebook.device)
    with profiler.region("vq_ema_apply", nbytes=4 * (5 * k * d + 3 * k), bound="hbm"):
        N.check(lib.smt_vq_ema_apply(N.ptr(codebook), N.ptr(k_sum), N.ptr(k_elem), N.ptr(stats), N.ptr(k_rand),
                                     float(mu), float(threshold), k, d, N.ptr(metrics), N.ptr(prep), prep.numel(),
                                     N.stream_ptr()), "smt_vq_ema_apply")
    return metrics, prep
