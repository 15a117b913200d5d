"""Cycles per frame in the phases of stft_loss_fwd_kernel<1024, 64> from the -DSMT_FFT_STAMP=1 build (tools/fft_phases.sh):
load + window | pass 1 (radix 16) | pass 2 (radix 16, twiddles) | pass 3 (radix 4) | spectra split + sums | whole frame."""
import ctypes, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import native, spectral

g = torch.Generator(device="cuda").manual_seed(0)
y = torch.randn(32, 145408, device="cuda", generator=g) * 0.1
yh = y + 0.05 * torch.randn(32, 145408, device="cuda", generator=g)
lib = ctypes.CDLL(native.LIB_PATH)
buf = (ctypes.c_ulonglong * (8192 * 8))()
spectral.stft_loss(y, yh, None, 1024, 120, 600, True)
torch.cuda.synchronize()
lib.smt_fft_debug_dump(buf, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
spectral.stft_loss(y, yh, None, 1024, 120, 600, True)
b.record()
torch.cuda.synchronize()
lib.smt_fft_debug_dump(buf, 1)
import numpy as np
m = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
m = m[m[:, 6] > 0]
names = ["load", "pass1", "pass2", "pass3", "split+sums", "frame"]
print(f"waves sampled {len(m)}, launch {a.elapsed_time(b) * 1e3:.1f} us (stamped build)")
print("  ".join(f"{nm} {m[:, i].mean():8.0f}" for i, nm in enumerate(names)), "  (mean s_memtime ticks per frame)")
print("  ".join(f"{nm} {np.median(m[:, i]):8.0f}" for i, nm in enumerate(names)), "  (median)")
