"""Per-phase cycle sums of conv_ws2_kernel from the -DSMT_WS_STAMP=1 build (tools/ws_phases.sh): for waves 0-3 (multiply,
then write out) and waves 4-7 (write out the previous tile, then multiply) of every workgroup, cycles per tile spent in:
barrier wait | input-tile DMA issue | deferred epilogue | epilogue-operand DMA issue | MFMA loop | vmcnt(0) wait | epilogue."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import convops as C, native

B, T, dt = 32, int(os.environ.get("T", 72704)), torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
u1 = torch.randn(B, T, 512, device="cuda", generator=g).relu().to(dt)
u2 = torch.empty_like(u1)
dz = torch.randn(B, T, 512, device="cuda", generator=g).to(dt)
dh1 = torch.empty_like(u1)
dh2 = torch.randn(B, T, 128, device="cuda", generator=g).to(dt)
bias = torch.randn(128, device="cuda", generator=g)
lib = ctypes.CDLL(native.LIB_PATH)
names = ["barrier", "dma in", "epi(prev)", "dma epi", "mfma", "vmcnt", "epi", "tiles"]
for d_, (k, dil) in enumerate([(3, 1), (5, 3), (7, 9), (9, 27)]):
    pad = (k - 1) * dil // 2
    w = torch.randn(128, 128, k, device="cuda", generator=g) / (128 * k) ** 0.5
    sl = slice(128 * d_, 128 * (d_ + 1))
    wf, wb = C._pack_fwd(w, dt, True), C._pack_bwd(w, dt, True)
    for mode in ("fwd", "dgrad"):
        if mode == "fwd":
            d = C._base_desc(u1[:, :, sl], None, None, 128, 128, k, 1, dil, pad, T, t_y=T)
            d.w, d.bias = C._p(wf), C._p(bias)
            C._use_dma(d, wf)
            C._set_act_out(d, u2[:, :, sl], [C.dropout_key(7, d_)], 6554, 1.0 / 0.9, 128)
        else:
            d = C._dgrad_stride1(dh2, wb, dh1[:, :, sl], k, dil, pad)
            C._use_dma(d, wb)
            C._set_act_grad(d, u1[:, :, sl], 1.0 / 0.9)
            d.res, d.bs_res, d.ld_res = C._geom(dz[:, :, sl])
        for _ in range(3):
            C.N.check(C.N.lib().smt_conv1d_ntc(ctypes.byref(d), C.N.stream_ptr()), "conv")
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 64))()
        lib.smt_ws_debug_dump(buf, 256 * 64, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        C.N.check(C.N.lib().smt_conv1d_ntc(ctypes.byref(d), C.N.stream_ptr()), "conv")
        e.record(); torch.cuda.synchronize()
        lib.smt_ws_debug_dump(buf, 256 * 64, 1)
        a = np.array(buf, dtype=np.float64).reshape(256, 8, 8)
        tiles = a[:, :, 7]
        per = a[:, :, :7] / np.maximum(tiles[:, :, None], 1)
        print(f"k={k} {mode}: {s.elapsed_time(e) * 1e3:.1f} us, tiles per workgroup {tiles[:, 0].mean():.1f}; cycles per tile (mean over workgroups)")
        for grp, sel in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
            if tiles[:, sel].sum() == 0:
                continue
            m = per[:, sel, :].mean(axis=(0, 1))
            print(f"   {grp}: " + "  ".join(f"{n} {v:7.0f}" for n, v in zip(names[:7], m)) + f"   sum {m.sum():7.0f}")
