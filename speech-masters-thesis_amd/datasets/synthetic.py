"""LJSpeech-shaped synthetic clips behind the reference's dataset contract
(``Dataset(config, split)``, 7-slot items, static ``collate``; datasets/ljspeech.py:17-140).

There is no LJSpeech audio offline, so benchmarks and tests draw clips from the seeded
generator of SURVEY.md 8(d): 0.5*(0.6*sum_{h<=8} sin(2 pi h f0 t + phi_h)/h + 0.4*U(-1,1)),
f0 ~ U[90, 250] Hz, clamped to [-1, 1], 22.05 kHz.
"""
import math

import torch
from torch.utils.data import Dataset

from datasets.ljspeech import LJSpeech, TRUNC_MOD


def synth_clip(length, seed, sample_rate=22050):
    g = torch.Generator().manual_seed(seed)
    f0 = 90.0 + 160.0 * torch.rand(1, generator=g, dtype=torch.float64)
    phi = 2 * math.pi * torch.rand(8, generator=g, dtype=torch.float64)
    t = torch.arange(length, dtype=torch.float64) / sample_rate
    tone = sum(torch.sin(2 * math.pi * h * f0 * t + phi[h - 1]) / h for h in range(1, 9))
    noise = torch.rand(length, generator=g, dtype=torch.float64) * 2 - 1
    return (0.5 * (0.6 * tone + 0.4 * noise)).clamp(-1, 1).to(torch.float32)


class SyntheticLJSpeech(Dataset):
    def __init__(self, config, split):
        super().__init__()
        ds = config.dataset
        self.n = int(ds.get("num_clips", 256)) if split == "train" else 10
        self.offset = 0 if split == "val" else 10
        self.fixed_length = int(ds.get("clip_length", 145408))
        self.ragged = bool(ds.get("ragged", False))
        self.sample_rate = int(ds.get("sample_rate", 22050))
        self.use_audio = ds.get("use_audio", True)
        self.segment_length = int(ds.get("segment_length", -1))

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        seed = self.offset + index
        length = self.fixed_length
        if self.ragged:
            g = torch.Generator().manual_seed(10_000 + seed)
            length = int(torch.randint(24064, 222720 + 1, (1,), generator=g))
        audio = synth_clip(length, seed, self.sample_rate)
        if 0 < self.segment_length < audio.shape[-1]:
            audio = audio[:self.segment_length]
        audio = audio[:len(audio) - len(audio) % TRUNC_MOD]
        if not self.use_audio:
            return None, None, None, None, None, None, None
        return None, None, None, None, audio, audio.shape[-1], None

    collate = staticmethod(LJSpeech.collate)


class SyntheticTTS(Dataset):
    """Token ids + log-mel-like frames with LJSpeech-like proportions behind the reference's dataset contract, for the
    token-to-spectrogram models (GlowTTS): item = (token [Tx] int64, Tx, spect [n_mels, Ty] fp32, Ty, None, None, None)."""

    def __init__(self, config, split):
        super().__init__()
        ds = config.dataset
        self.n = int(ds.get("num_clips", 256)) if split == "train" else 10
        self.offset = 0 if split == "val" else 10
        self.max_tokens = int(ds.get("max_tokens", 160))
        self.fpt = int(ds.get("frames_per_token", 5))
        self.n_vocab = int(ds.get("n_vocab", 148)) + int(bool(ds.get("intersperse_blanks", False)))
        self.n_mels = int(ds.get("n_mels", 80))

    def __len__(self):
        return self.n

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(50_000 + self.offset + index)
        tx = int(torch.randint(max(2, self.max_tokens // 2), self.max_tokens + 1, (1,), generator=g))
        ty = tx * self.fpt + int(torch.randint(0, 2 * self.fpt, (1,), generator=g))
        token = torch.randint(1, self.n_vocab, (tx,), generator=g)
        # frames that depend on the token under them (so that an alignment exists to be found) + noise, log-mel range
        owner = torch.div(torch.arange(ty) * tx, ty, rounding_mode="floor").clamp(max=tx - 1)
        basis = torch.randn(self.n_vocab, self.n_mels, generator=torch.Generator().manual_seed(7))
        spect = (basis[token[owner]] * 1.5 - 4.0 + 0.5 * torch.randn(ty, self.n_mels, generator=g)).t().contiguous()
        return token, tx, spect, ty, None, None, None

    collate = staticmethod(LJSpeech.collate)
