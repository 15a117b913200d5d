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
B, T = 32, 72704
dt = torch.bfloat16
x = torch.randn(B, T, 128, device="cuda").to(dt); y = torch.empty_like(x)
bias = torch.randn(128, device="cuda")
for (k, dil) in [(1, 1), (9, 27)]:
    w = torch.randn(128, 128, k, device="cuda") / (128 * k) ** 0.5
    wp = C._pack_fwd(w, dt); pad = (k - 1) * dil // 2
    def run():
        d = C._base_desc(x, y, None, 128, 128, k, 1, dil, pad, T)
        d.w, d.bias = C._p(wp), C._p(bias)
        C._launch(d, "x")
    print(f"dbg={os.environ.get('SMT_CONV_DBG','0')} k={k} dil={dil}: {timeit(run):9.1f} us")
