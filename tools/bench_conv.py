"""Micro-benchmark of the conv GEMM kernel at the largest level's shape (B=32, T=72704)."""
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
big = torch.randn(B, T, 512, device="cuda").to(dt)
big2 = torch.empty_like(big)
cont_in = torch.randn(B, T, 128, device="cuda").to(dt)
cont_out = torch.empty_like(cont_in)
res = torch.randn(B, T, 128, device="cuda").to(dt)
bias = torch.randn(128, device="cuda")
for (k, dil) in [(1, 1), (3, 1), (9, 27)]:
    w = torch.randn(128, 128, k, device="cuda") / (128 * k) ** 0.5
    wp = C._pack_fwd(w, dt)
    pad = (k - 1) * dil // 2
    def run(x, y, r=None, act=None):
        d = C._base_desc(x, y, None, 128, 128, k, 1, dil, pad, T)
        d.w, d.bias = C._p(wp), C._p(bias)
        if r is not None:
            d.res, d.bs_res, d.ld_res = C._geom(r)
        if act is not None:
            C._set_act_out(d, act, [123], 6554, 1.111, 128)
        C._launch(d, "x")
    flops = 2.0 * B * T * 128 * 128 * k
    for name, fn in [
        ("contig in/out", lambda: run(cont_in, cont_out)),
        ("slice in (ld 512), contig out", lambda: run(big[:, :, 128:256], cont_out)),
        ("slice in, slice out", lambda: run(big[:, :, 128:256], big2[:, :, 128:256])),
        ("contig + residual", lambda: run(cont_in, cont_out, res)),
        ("contig + act_out only", lambda: run(cont_in, None if False else cont_out, None, res)),
    ]:
        us = timeit(fn)
        print(f"k={k} dil={dil:2d} {name:32s} {us:9.1f} us  {flops / us / 1e6:7.1f} TF")

print("---- wgrad")
for (k, dil) in [(1, 1), (3, 1), (5, 3), (9, 27)]:
    pad = (k - 1) * dil // 2
    dw = torch.empty(128, 128, k, device="cuda"); db = torch.empty(128, device="cuda")
    def runw(x, dy):
        d = C._base_desc(x, dy, None, 128, 128, k, 1, dil, pad, T)
        C._wgrad(d, dw, 128 * k, k, 1, list(range(k)), db)
    flops = 2.0 * B * T * 128 * 128 * k
    us = timeit(lambda: runw(cont_in, res))
    print(f"wgrad k={k} dil={dil:2d} contig {us:9.1f} us  {flops / us / 1e6:7.1f} TF")
    us = timeit(lambda: runw(big[:, :, 128:256], big2[:, :, 0:128]))
    print(f"wgrad k={k} dil={dil:2d} sliced {us:9.1f} us  {flops / us / 1e6:7.1f} TF")
