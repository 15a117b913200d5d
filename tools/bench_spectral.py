"""Stand-alone timing of the spectral kernels at the bench shapes (batch 32 x 145,408 samples): the three loss resolutions
forward and backward, and the log-mel front end.  HIP events over `iters` launches each; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (the backward entry is two kernels)."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "speech-masters-thesis_amd"))
from smt_amd import spectral  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--t", type=int, default=145408)
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.randn(args.batch, args.t, device="cuda", generator=g) * 0.1
    yh = (y + 0.05 * torch.randn(args.batch, args.t, device="cuda", generator=g)).requires_grad_(True)
    for n_fft, hop, win in ((2048, 240, 1200), (1024, 120, 600), (512, 50, 240)):
        def fwd():
            return spectral.stft_loss(y, yh, None, n_fft, hop, win, True)
        loss = fwd()

        def bwd():
            torch.autograd.grad(loss, yh, retain_graph=True)
        print(f"stft_loss n_fft={n_fft} hop={hop}: fwd {timed(fwd, args.iters):8.1f} us   bwd {timed(bwd, args.iters):8.1f} us"
              f"   frames {spectral.num_frames(args.t, n_fft, hop) * args.batch}", flush=True)
    # front end (configs: n_fft 1024, hop 256, win 1024, 80 mels)
    from datasets.transforms import MelSpectrogram
    mel = MelSpectrogram(n_fft=1024, hop_length=256, win_length=1024, n_mels=80, sample_rate=22050, f_min=0.0, f_max=8000.0).cuda()
    us = timed(lambda: spectral.log_mel(y, mel.mel_basis, mel.band, 1024, 256, 1024), args.iters)
    print(f"melspec 1024/256: {us:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
