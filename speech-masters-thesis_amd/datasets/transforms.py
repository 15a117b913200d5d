"""Audio front end: windowed STFT magnitude and log-mel spectrogram modules with the
reference's constructor signatures (datasets/transforms.py:16-123).

Semantics: periodic Hann of ``win_length`` centre-padded to ``n_fft``; reflect padding of
``(n_fft - hop) // 2`` samples per side; frames every ``hop``; bins 0..n_fft/2; magnitude
(not power); mel = Slaney filterbank @ magnitude, then log(clamp(., 1e-5)).
``STFT.inverse`` is never called in the reference and is not provided.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from smt_amd import spectral
from utils.torch_utils import safe_log


def slaney_mel_filterbank(sample_rate, n_fft, n_mels, f_min=0.0, f_max=None):
    """Triangular filters on the Slaney mel scale with area normalisation -- the documented
    default of librosa.filters.mel, which the reference calls (transforms.py:38-44)."""
    f_max = sample_rate / 2.0 if f_max is None else f_max
    lin_step, knee_hz, log_step = 200.0 / 3.0, 1000.0, math.log(6.4) / 27.0
    knee_mel = knee_hz / lin_step

    def to_mel(hz):
        return hz / lin_step if hz < knee_hz else knee_mel + math.log(hz / knee_hz) / log_step

    def to_hz(mel):
        return lin_step * mel if mel < knee_mel else knee_hz * math.exp(log_step * (mel - knee_mel))

    edges = np.array([to_hz(m) for m in np.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)])
    bins = np.linspace(0.0, sample_rate / 2.0, n_fft // 2 + 1)
    bank = np.zeros((n_mels, bins.size))
    for m in range(n_mels):
        rise = (bins - edges[m]) / (edges[m + 1] - edges[m])
        fall = (edges[m + 2] - bins) / (edges[m + 2] - edges[m + 1])
        bank[m] = np.clip(np.minimum(rise, fall), 0.0, None) * (2.0 / (edges[m + 2] - edges[m]))
    return bank.astype(np.float32)


class STFT(nn.Module):
    def __init__(self, n_fft: int = 1024, hop_length: int = 256, win_length: int = None, window: str = "hann"):
        super().__init__()
        assert window == "hann", "only the Hann window is used by the reference configs"
        self.n_fft, self.hop_length = n_fft, hop_length
        self.win_length = win_length if win_length else n_fft
        assert n_fft >= self.win_length
        self.window = window
        self.pad_amount = (n_fft - hop_length) // 2

    def num_frames(self, num_samples):
        return (num_samples + 2 * self.pad_amount - self.n_fft) // self.hop_length + 1

    def forward(self, input_data):
        """[B, T] or [B, 1, T] -> magnitudes [B, n_fft/2 + 1, frames]."""
        b, t = input_data.shape[0], input_data.shape[-1]
        return spectral.stft_magnitude(input_data.reshape(b, t), self.n_fft, self.hop_length, self.win_length)

    def inverse(self, magnitude, phase):
        """magnitude, phase [B, n_fft/2 + 1, frames] -> [B, 1, T'] (transforms.py:125-156): inverse FFT per frame, synthesis
        window, overlap-add, window-sum-square normalisation, pad_amount trimmed on both sides -- one HIP kernel pair."""
        return spectral.stft_inverse(magnitude, phase, self.n_fft, self.hop_length, self.win_length)


class MelSpectrogram(nn.Module):
    def __init__(self, n_fft=1024, hop_length=256, win_length=None, n_mels=80, sample_rate=22050, f_min=0.0,
                 f_max=None):
        super().__init__()
        self.n_fft, self.hop_length = n_fft, hop_length
        self.stft = STFT(n_fft=n_fft, hop_length=hop_length, win_length=win_length or n_fft, window="hann")
        basis = slaney_mel_filterbank(sample_rate, n_fft, n_mels, f_min, f_max)
        self.register_buffer("mel_basis", torch.from_numpy(basis))
        nz = basis > 0
        lo = np.where(nz.any(1), nz.argmax(1), 0)
        hi = np.where(nz.any(1), basis.shape[1] - nz[:, ::-1].argmax(1), 0)
        self.register_buffer("band", torch.from_numpy(np.stack([lo, hi], 1).astype(np.int32)), persistent=False)

    def forward(self, audio):
        assert audio.min() >= -1 and audio.max() <= 1
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        stft = self.stft
        return spectral.log_mel(audio, self.mel_basis, self.band, stft.n_fft, stft.hop_length, stft.win_length)

    def mel_len(self, audio_len):
        return audio_len // self.hop_length
