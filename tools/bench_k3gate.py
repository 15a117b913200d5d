"""Micro-benchmark of smt_conv_k3gate_fwd at the top level of the bench model (B = 32, T = 72,704, width 64)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import convops as C
b, t = 32, 72704
g = torch.Generator(device="cuda").manual_seed(0)
u2 = torch.randn(b, t, 512, device="cuda", generator=g).to(torch.bfloat16)
x = torch.randn(b, t, 64, device="cuda", generator=g).to(torch.bfloat16)
w3 = [torch.randn(128, 128, 1, device="cuda", generator=g) / 11 for _ in range(4)]
w1 = [torch.randn(128, 64, 1, device="cuda", generator=g) / 8 for _ in range(4)]
b3 = [torch.randn(128, device="cuda", generator=g) for _ in range(4)]
b1 = [torch.randn(128, device="cuda", generator=g) for _ in range(4)]
z = torch.empty_like(u2); gg = torch.empty_like(x)
w3p, w1p = C._pack_cat_fwd_swz(w3, torch.bfloat16), C._pack_cat_fwd(w1, torch.bfloat16)
b3c, b1c = C._cat_bias(b3), C._cat_bias(b1)
fn = lambda: C._conv_k3gate(u2, x, w3p, w1p, b3c, b1c, z, gg, None)
for _ in range(3): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): fn()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 10 * 1e3
nbytes = b * t * (1024 + 128) * 2
print(f"conv_k3gate B={b} T={t}: {us:.0f} us, {nbytes / us / 1e6:.2f} TB/s of the algorithmic 2.3 KB/row")
