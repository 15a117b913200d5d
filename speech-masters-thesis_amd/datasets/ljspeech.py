"""LJSpeech-1.1 behind the reference's dataset contract (datasets/ljspeech.py:17-140).

Items are 7-tuples ``(token, token_len, spect, spect_len, audio, audio_len, speaker)`` with
``None`` in unused slots; ``collate`` zero-pads to the batch maximum.  LJSpeech wavs are
16-bit PCM at 22,050 Hz, so ``int16 / 32768`` read with the stdlib ``wave`` module equals what
``librosa.load`` returns without resampling.  The phoneme front end (CMUDict parser) is outside
the VQ-VAE path: ``use_token`` is forced off by ``get_model`` for reconstruction models and is
rejected here.
"""
import math
import os
import random
import wave

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

TRUNC_MOD = 512  # clips are cut to a multiple of this (ljspeech.py:14, :82)


def read_wav(path):
    with wave.open(path, "rb") as f:
        assert f.getsampwidth() == 2 and f.getnchannels() == 1, "LJSpeech is 16-bit mono PCM"
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2")
    return torch.from_numpy(pcm.astype(np.float32) / 32768.0)


def _pad_stack(items, lengths, value=0.0):
    longest = int(lengths.max())
    return torch.stack([F.pad(x, (0, longest - x.shape[-1]), value=value) for x in items], dim=0)


class LJSpeech(Dataset):
    def __init__(self, config, split: str):
        super().__init__()
        ds = config.dataset
        self.root = ds.dataset_path
        self.segment_length = ds.segment_length
        if self.segment_length > 0:
            assert self.segment_length % TRUNC_MOD == 0, \
                f"segment_length={self.segment_length} must be a multiple of {TRUNC_MOD}"
        self.use_token, self.use_spect, self.use_audio = ds.use_token, ds.use_spect, ds.use_audio
        with open(os.path.join(self.root, "metadata.csv"), encoding="utf-8") as f:
            ids = [line.strip().split("|")[0] for line in f if line.strip()]
        paths = [os.path.join(self.root, "wavs", f"{i}.wav") for i in ids]
        if split == "train":       # fixed split: first ten clips validate (ljspeech.py:40-45)
            self.audio = paths[10:]
        elif split == "val":
            self.audio = paths[:10]
        else:
            raise ValueError(f"LJSpeech not implemented for split {split}")
        self._mel = None
        self._mel_args = dict(sample_rate=ds.sample_rate, n_fft=ds.n_fft, win_length=ds.win_length,
                              hop_length=ds.hop_length, n_mels=ds.n_mels, f_min=0.0, f_max=8000.0)

    def __len__(self):
        return len(self.audio)

    def __getitem__(self, index):
        audio = read_wav(self.audio[index])
        if 0 < self.segment_length < audio.shape[-1]:
            start = random.randint(0, audio.shape[-1] - self.segment_length)
            audio = audio[start:start + self.segment_length]
        audio = audio[:len(audio) - len(audio) % TRUNC_MOD]
        audio_len = audio.shape[-1]
        spect = spect_len = None
        if self.use_spect:
            raise NotImplementedError("use_spect: log-mel is computed on the device by datasets.transforms."
                                      "MelSpectrogram, not inside DataLoader workers")
        if self.use_token:
            raise NotImplementedError("use_token: the CMUDict text front end is outside the VQ-VAE hot path")
        if not self.use_audio:
            audio = audio_len = None
        return None, None, spect, spect_len, audio, audio_len, None

    @staticmethod
    def collate(batch):
        token, token_len, spect, spect_len, audio, audio_len, _ = zip(*batch)
        out = [None] * 7
        if token[0] is not None:
            out[1] = torch.tensor(token_len, dtype=torch.long)
            out[0] = _pad_stack(token, out[1])
        if spect[0] is not None:
            out[3] = torch.tensor(spect_len, dtype=torch.long)
            out[2] = _pad_stack(spect, out[3], value=math.log(1e-7))
        if audio[0] is not None:
            out[5] = torch.tensor(audio_len, dtype=torch.long)
            out[4] = _pad_stack(audio, out[5]).unsqueeze(1)
        return tuple(out)
