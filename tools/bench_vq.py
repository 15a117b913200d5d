"""Micro-benchmark of the VQ kernels (HIP events on the launch stream)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import vq  # noqa: E402


def timeit(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def encoder_like(n, k, d, seed=7):
    """Rows sitting on top of their code with a large common offset (what an untrained encoder emits): many near-ties."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(n, d, generator=g) * 0.3 + 3.0 * torch.randn(1, d, generator=g)
    rows = base + torch.randn(n, d, generator=g) * 1e-4
    cb = rows[torch.randperm(n, generator=g)][:k].clone()
    return base.cuda(), cb.cuda()


CASES = [(4544, 256, "gauss"), (36352, 256, "gauss"), (36352, 1024, "gauss"), (36352, 1024, "encoder-like")]
if len(sys.argv) > 1:                       # e.g. `bench_vq.py 2` = only the third case (for rocprofv3 runs)
    CASES = [CASES[int(a)] for a in sys.argv[1:]]
for n, k, kind in CASES:
    d = 128
    if kind == "gauss":
        x = torch.randn(n, d, device="cuda")
        cb = torch.randn(k, d, device="cuda")
    else:
        x, cb = encoder_like(n, k, d)
    prep = vq.prepare(cb)
    us = timeit(lambda: vq.vq_forward_raw(x, cb, prep=prep))
    us_cold = timeit(lambda: vq.vq_forward_raw(x, cb))
    alg_bytes = n * (4 * d + 8 + 4 * d + 4) + 4 * k * d
    flops = 2.0 * n * k * d
    print(f"vq_forward N={n} K={k} {kind}: {us:8.1f} us with cached prep ({us_cold:.1f} without)  {alg_bytes / us / 1e6:7.3f} TB/s alg  "
          f"{3 * flops / us / 1e6:7.2f} TFLOP/s bf16-MFMA  queued={int(vq.vq_forward_raw(x, cb, prep=prep)[3][3].item())}")
    idx = vq.vq_forward_raw(x, cb, prep=prep)[0]
    stats = torch.empty(vq.ema_stats_numel(k, d), device="cuda")
    us = timeit(lambda: vq.ema_accumulate(x, idx, None, k, stats))
    print(f"ema_accumulate N={n} K={k}: {us:8.1f} us  {n * (4 * d + 8) / us / 1e6:7.3f} TB/s alg")
    ks, ke = cb.clone(), torch.ones(k, device="cuda")
    us = timeit(lambda: vq.ema_apply(cb.clone(), ks, ke, stats, cb, 0.99, 1.0, prep))
    print(f"ema_apply (+ prep refresh, + a clone) K={k}: {us:8.1f} us")
