"""Fused 1x1 backward (smt_conv1x1_bwd) vs the two kernels it replaces, largest level (B=32, T=72704)."""
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
B, T, c = 32, int(os.environ.get("T", 72704)), 128
dt = torch.bfloat16
dzb = torch.randn(B, T, 512, device="cuda").to(dt); ub = torch.relu(torch.randn(B, T, 512, device="cuda")).to(dt)
dz, u2 = dzb[:, :, 128:256], ub[:, :, 128:256]
dx = torch.empty(B, T, c, device="cuda", dtype=dt)
w = torch.randn(c, c, 1, device="cuda") / c ** 0.5
wb = C._pack_bwd(w, dt, True)
dw, db = torch.empty_like(w), torch.empty(c, device="cuda")
def desc():
    d = C._dgrad_stride1(dz, wb, dx, 1, 1, 0); C._use_dma(d, wb); C._set_act_grad(d, u2, 1.111); return d
t_d = timeit(lambda: C._launch(desc(), "t"))
t_w = timeit(lambda: C._wgrad(C._base_desc(u2, dz, None, c, c, 1, 1, 1, 0, T), dw, c, 1, 1, [0], db))
t_f = timeit(lambda: C._conv1x1_bwd(desc(), dw, c, 1, db))
gb = B * T * 768 / 1e9
print(f"dgrad {t_d:7.1f} us  wgrad {t_w:7.1f} us  sum {t_d + t_w:7.1f} us | fused {t_f:7.1f} us ({gb / t_f * 1e3:5.2f} TB/s of 768 B/row)")
