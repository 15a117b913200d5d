"""Fused K1 backward (smt_conv_k1_bwd) at the largest level (B=32, T=72704)."""
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
dh = torch.randn(B, T, 512, device="cuda").to(dt); x = torch.randn(B, T, 64, device="cuda").to(dt)
dout = torch.randn(B, T, 64, device="cuda").to(dt); dx = torch.empty_like(x)
w = torch.randn(512, 64, 1, device="cuda") / 8
wb = C._pack_bwd(w, dt)
dw, db = torch.empty_like(w), torch.empty(512, device="cuda")
t = timeit(lambda: C._conv_k1_bwd(dh, x, wb, dout, dx, None, dw, db))
print(f"k1_bwd {t:7.1f} us  {B * T * (512 + 192) * 2 / t / 1e6:5.2f} TB/s")
