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


for n, k in [(4544, 256), (36352, 256), (36352, 1024)]:
    d = 128
    x = torch.randn(n, d, device="cuda")
    cb = torch.randn(k, d, device="cuda")
    us = timeit(lambda: vq.vq_forward_raw(x, cb))
    alg_bytes = n * (4 * d + 8 + 4 * d + 4) + 4 * k * d
    flops = 2.0 * n * k * d
    print(f"vq_forward N={n} K={k}: {us:8.1f} us  {alg_bytes / us / 1e6:7.3f} TB/s alg  {flops / us / 1e6:7.2f} TFLOP/s"
          f"  ambiguous={int(vq.vq_forward_raw(x, cb)[3][3].item())}")
    idx = vq.vq_forward_raw(x, cb)[0]
    stats = torch.empty(vq.ema_stats_numel(k, d), device="cuda")
    us = timeit(lambda: vq.ema_accumulate(x, idx, None, k, stats))
    print(f"ema_accumulate N={n} K={k}: {us:8.1f} us  {n * (4 * d + 8) / us / 1e6:7.3f} TB/s alg")
