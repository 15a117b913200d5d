"""K1 activated-output kernel (64 -> 512) at the largest level (B=32, T=72704)."""
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
x = torch.randn(B, T, 64, device="cuda").to(dt); u = torch.empty(B, T, 512, device="cuda", dtype=dt)
w = torch.randn(512, 64, 1, device="cuda") / 8; bias = torch.randn(512, device="cuda")
wp = C._pack_fwd(w, dt)
def run():
    d = C._base_desc(x, None, None, 64, 512, 1, 1, 1, 0, T, t_y=T)
    d.w, d.bias = C._p(wp), C._p(bias)
    C._set_act_out(d, u, [1, 2, 3, 4], 6554, 1.111, 128)
    d.zero_page = C._p(C._zero_page(x.device))
    C._launch(d, "t")
t = timeit(run)
print(f"k1act {t:7.1f} us  {B * T * (64 + 512) * 2 / t / 1e6:5.2f} TB/s")
