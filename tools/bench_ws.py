"""Micro-benchmark of the K2 convs of a GatedHiFi block (128 -> 128, k = 3/5/7/9, dilation 1/3/9/27) exactly as the block
launches them: forward = activated output only (relu + counter dropout), data gradient = activation-gradient mask + residual;
operands are 128-channel slices of 512-channel tensors (row pitch 1 KiB).  T = rows per batch item (default: the top level).

    python tools/bench_ws.py            # SMT_CONV_NO_WS2=1 / SMT_CONV_NO_PIPE=1 select the older kernels for A/B runs
"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import convops as C


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


B, T = 32, int(os.environ.get("T", 72704))
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
u1 = torch.randn(B, T, 512, device="cuda", generator=g).relu().to(dt)
u2 = torch.empty_like(u1)
dz = torch.randn(B, T, 512, device="cuda", generator=g).to(dt)
dh1 = torch.empty_like(u1)
dh2 = torch.randn(B, T, 128, device="cuda", generator=g).to(dt)
bias = torch.randn(128, device="cuda", generator=g)
lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
for d_, (k, dil) in enumerate([(3, 1), (5, 3), (7, 9), (9, 27)]):
    pad = (k - 1) * dil // 2
    w = torch.randn(128, 128, k, device="cuda", generator=g) / (128 * k) ** 0.5
    sl = slice(128 * d_, 128 * (d_ + 1))
    wf, wb = C._pack_fwd(w, dt, True), C._pack_bwd(w, dt, True)

    def fwd():
        d = C._base_desc(u1[:, :, sl], None, None, 128, 128, k, 1, dil, pad, T, t_y=T)
        d.w, d.bias = C._p(wf), C._p(bias)
        C._use_dma(d, wf)
        C._set_act_out(d, u2[:, :, sl], [C.dropout_key(7, d_)], 6554, 1.0 / 0.9, 128)
        return d

    def dgrad():
        d = C._dgrad_stride1(dh2, wb, dh1[:, :, sl], k, dil, pad)
        C._use_dma(d, wb)
        C._set_act_grad(d, u1[:, :, sl], 1.0 / 0.9)
        d.res, d.bs_res, d.ld_res = C._geom(dz[:, :, sl])
        return d

    flops = 2.0 * B * T * 128 * 128 * k
    for name, mk in (("fwd", fwd), ("dgrad", dgrad)):
        desc = mk()
        kern = C._kernel_of(desc)
        us = timeit(lambda: C.N.check(C.N.lib().smt_conv1d_ntc(C.ctypes.byref(desc), C.N.stream_ptr()), "conv"))
        print(f"k={k} dil={dil:2d} {name:5s} {kern:13s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF  ({flops / us / 1e6 / 2500:.3f} of peak)")
