"""Micro-benchmark of the conv weight-gradient kernels at the largest level's shape (B=32, T=72704)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import convops as C

def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

B, T = 32, int(os.environ.get("T", 72704))
dt = torch.bfloat16
x = torch.randn(B, T, 128, device="cuda").to(dt)
dy = torch.randn(B, T, 128, device="cuda").to(dt)
for (k, dil) in [(1, 1), (3, 1), (5, 3), (7, 9), (9, 27)]:
    pad = (k - 1) * dil // 2
    dw = torch.empty(128, 128, k, device="cuda"); db = torch.empty(128, device="cuda")
    def runw():
        d = C._base_desc(x, dy, None, 128, 128, k, 1, dil, pad, T)
        C._wgrad(d, dw, 128 * k, k, 1, list(range(k)), db)
    flops = 2.0 * B * T * 128 * 128 * k
    us = timeit(runw)
    print(f"wgrad k={k} dil={dil:2d} {us:9.1f} us  {flops / us / 1e6:7.1f} TF  {2 * B * T * 256 / us / 1e6:6.2f} TB/s (x + dy once)")
