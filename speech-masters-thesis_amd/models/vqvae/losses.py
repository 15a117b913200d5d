"""Training losses of the VQ-VAE (reference models/vqvae/losses.py).

Signals are [B, T]; masks are prefix masks given by per-item lengths.
  MultiResolutionSpectralLoss (losses.py:11-55): per resolution, Frobenius norm over
    (bins, frames) of the masked magnitude difference, averaged over the batch, plus the
    same on log magnitudes when ``log``; divided by the number of resolutions.
  MultiNormReconstructionLoss (losses.py:58-80): l1*L1 + l2*MSE + linf * sum_j mean_B
    top-k_j((y - yh)^2) on masked signals.
"""
from typing import Iterable

import torch
import torch.nn as nn

from datasets.transforms import STFT
from smt_amd import spectral


class MultiResolutionSpectralLoss(nn.Module):
    def __init__(self, n_ffts: Iterable[int], hop_lengths: Iterable[int], win_lengths: Iterable[int] = None,
                 window: str = "hann", log: bool = False):
        super().__init__()
        win_lengths = n_ffts if win_lengths is None else win_lengths
        assert len(n_ffts) == len(hop_lengths) == len(win_lengths)
        self.stfts = nn.ModuleList(STFT(n_fft=n, hop_length=h, win_length=w, window=window)
                                   for n, h, w in zip(n_ffts, hop_lengths, win_lengths))
        self.log = log

    @staticmethod
    def frame_mask(lens, stft, frames):
        """Frame f is kept iff the sample under its centre tap is unmasked: the reference pads the
        mask with ones (left) / zeros (right) and slices [n_fft//2 : -n_fft//2+1 : hop] (losses.py:33-37)."""
        centre = stft.n_fft // 2 - stft.pad_amount + stft.hop_length * torch.arange(frames, device=lens.device)
        return (centre[None, :] < lens[:, None]).to(torch.float32)

    def forward(self, y, yh, lens):
        loss = 0.0
        for stft in self.stfts:
            loss = loss + spectral.stft_loss(y, yh, lens, stft.n_fft, stft.hop_length, stft.win_length, self.log)
        return loss / len(self.stfts)


class MultiNormReconstructionLoss(nn.Module):
    def __init__(self, l1: float = 0.0, l2: float = 1.0, linf: float = 0.02, linf_topk: int = 2048):
        super().__init__()
        self.l1, self.l2, self.linf, self.linf_topk = l1, l2, linf, linf_topk

    def forward(self, y, yh, lens):
        """One radix-select kernel per call instead of materialised d^2 + torch.topk (smt_amd.spectral.recon_loss)."""
        if not self.linf:        # no top-k term: the kernel still needs a valid k
            return spectral.recon_loss(y, yh, lens, self.l1, self.l2, 0.0, 1)
        return spectral.recon_loss(y, yh, lens, self.l1, self.l2, self.linf, self.linf_topk)
