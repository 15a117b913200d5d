"""Phase timing of vq_search_kernel from an ablation build with -DVQ_ABL=16 (tools/ablate_vq.sh): per-workgroup
timestamps (100 MHz wall clock) -> start spread, prologue, main loop, merge, epilogue."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import vq, native
n, k, d = 36352, 1024, 128
x, cb = torch.randn(n, d, device="cuda"), torch.randn(k, d, device="cuda")
prep = vq.prepare(cb)
for _ in range(5):
    vq.vq_forward_raw(x, cb, prep=prep)
torch.cuda.synchronize()
groups = (n + 31) // 32
rgs = min(5, (groups + 255) // 256)
wgs = (groups + rgs - 1) // rgs
buf = (ctypes.c_longlong * (wgs * 6))()
lib = ctypes.CDLL(native.LIB_PATH)
assert lib.smt_vq_debug_dump(buf, wgs * 6) == 0
t = np.array(buf, dtype=np.int64).reshape(wgs, 6)[:, :5].astype(np.float64) * 0.01   # us
t0 = t[:, 0].min()
print(f"workgroups {wgs}; first start 0, last start {t[:, 0].max() - t0:.1f} us, last end {t[:, 4].max() - t0:.1f} us")
names = ["prologue (stage 0/1 + x rows)", "main loop", "merge + ambiguity", "epilogue"]
for i, nm in enumerate(names):
    dt = t[:, i + 1] - t[:, i]
    print(f"  {nm:32s} mean {dt.mean():6.2f}  min {dt.min():6.2f}  max {dt.max():6.2f} us")
late = t[:, 0] - t0 > 5
print(f"  workgroups starting later than 5 us: {late.sum()}; their mean total {(t[late, 4] - t[late, 0]).mean() if late.any() else 0:.1f} us; "
      f"others' mean total {(t[~late, 4] - t[~late, 0]).mean():.1f} us")
