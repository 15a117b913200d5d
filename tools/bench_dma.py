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
x = torch.randn(B, T, 128, device="cuda").to(dt); y = torch.empty_like(x); res = torch.randn(B, T, 128, device="cuda").to(dt)
u = torch.empty_like(x)
bias = torch.randn(128, device="cuda")
for (k, dil) in [(1, 1), (3, 1), (5, 3), (7, 9), (9, 27)]:
    w = torch.randn(128, 128, k, device="cuda") / (128 * k) ** 0.5
    pad = (k - 1) * dil // 2
    for dma in (False, True):
        wp = C._pack_fwd(w, dt, dma)
        def run(r=None, act=False, yy=y):
            d = C._base_desc(x, yy, None, 128, 128, k, 1, dil, pad, T, t_y=T)
            d.w, d.bias = C._p(wp), C._p(bias)
            if dma: C._use_dma(d, wp)
            if r is not None: d.res, d.bs_res, d.ld_res = C._geom(r)
            if act: C._set_act_out(d, u, [123], 6554, 1.111, 128)
            if r is not None and act is None: C._set_act_grad(d, res, 1.111)
            C._launch(d, "x")
        flops = 2.0 * B * T * 128 * 128 * k
        t1 = timeit(lambda: run()); t2 = timeit(lambda: run(res)); t3 = timeit(lambda: run(None, True, None)); t4 = timeit(lambda: run(res, None))
        gb = B * T * 128 * 2 / 1e9
        print(f"k={k} dil={dil:2d} dma={int(dma)}  plain {t1:7.1f} us ({flops/t1/1e6:6.1f} TF, {2*gb/t1*1e3:5.2f} TB/s)  +res {t2:7.1f} us ({3*gb/t2*1e3:5.2f} TB/s)  act-only {t3:7.1f} us  res+actgrad {t4:7.1f} us")
