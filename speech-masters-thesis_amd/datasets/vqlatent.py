"""VQ-Latent dataset (reference datasets/vqlatent.py:16-142): per-utterance files ``{"x": [float...], "q": [int...]}``
written by ``scripts/generate_vq_dataset.py`` plus ``metadata.json {compression_factor, vocab_size}``.

Token ids are shifted by ``OFFSET`` so that 0 / 1 stay free for ``<pad>`` / ``<bos>`` (vqlatent.py:84-88).  Items are the
7-slot tuples of the dataset contract; ``collate`` pads tokens with PAD, audio with zeros.

The files are Python pickles of plain lists, which is what downstream reference tooling reads; they are LOADED here
with an unpickler that refuses every global (no code can run from a file, whoever wrote it).
"""
import io
import json
import math
import os
import pickle
import random
from itertools import groupby

import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


class _PlainUnpickler(pickle.Unpickler):
    """dict / list / str / int / float / bool / None only: any GLOBAL opcode is an error."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"VQ-Latent files hold plain containers only; refusing {module}.{name}")


def load_plain_pickle(path):
    with open(path, "rb") as f:
        return _PlainUnpickler(io.BytesIO(f.read())).load()


def dump_plain_pickle(obj, path):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


class VQLatent(Dataset):

    PAD = 0  # <pad> token
    BOS = 1  # <bos> token
    OFFSET = 2  # number of special tokens the original vocabulary is shifted by

    def __init__(self, config, split: str):
        super().__init__()
        ds = config.dataset
        self.split = split
        self.dataset_path = ds.dataset_path
        self.pkl_files = sorted(f for f in os.listdir(os.path.join(self.dataset_path, split)) if f.endswith(".pkl"))
        with open(os.path.join(self.dataset_path, "metadata.json"), "r", encoding="utf-8") as f:
            self.metadata = json.load(f)
        self.segment_length = ds.segment_length
        self.remove_consecutive = ds.get("remove_consecutive", False)
        vocab = config.model.get("vocab_size", None) if config.get("model", None) is not None else None
        assert vocab is None or vocab == self.metadata["vocab_size"], \
            "Need to specify correct model vocab size for this dataset"
        self.use_token, self.use_spect, self.use_audio = ds.use_token, ds.use_spect, ds.use_audio

    def __len__(self):
        return len(self.pkl_files)

    def __getitem__(self, index):
        pkl = load_plain_pickle(os.path.join(self.dataset_path, self.split, self.pkl_files[index]))
        audio, token = pkl["x"], pkl["q"]
        speaker = torch.tensor((pkl["speaker"],), dtype=torch.long) if "speaker" in pkl else None
        if self.remove_consecutive:
            token = [t[0] for t in groupby(token)]
        cf = self.metadata["compression_factor"]
        if self.segment_length > 0 and len(token) > self.segment_length:
            start = random.randint(0, len(token) - self.segment_length)
            token = token[start:start + self.segment_length]
            audio = audio[start * cf:start * cf + self.segment_length * cf]      # audio is cf times longer
        # <bos> in front; every code shifted by OFFSET (BOS - OFFSET + OFFSET == BOS)
        token = torch.tensor([VQLatent.BOS - VQLatent.OFFSET] + list(token), dtype=torch.long).flatten() + VQLatent.OFFSET
        audio = torch.tensor(audio, dtype=torch.float32).flatten()
        token_len, audio_len = token.shape[-1], audio.shape[-1]
        if self.segment_length > 0:       # short examples are padded to the segment (vqlatent.py:95-98)
            token = F.pad(token, (0, self.segment_length + 2 - len(token)), mode="constant", value=VQLatent.PAD)
            audio = F.pad(audio, (0, self.segment_length * cf - len(audio)))
        spect = spect_len = None
        if self.use_spect:
            raise NotImplementedError("use_spect: log-mel is computed on the device by datasets.transforms."
                                      "MelSpectrogram, not inside DataLoader workers")
        if not self.use_audio:
            audio = audio_len = None
        if not self.use_token:
            token = token_len = None
        return token, token_len, spect, spect_len, audio, audio_len, speaker

    @staticmethod
    def collate(batch):
        """None entries are slots the config does not need (vqlatent.py:114-142)."""
        token, token_len, spect, spect_len, audio, audio_len, speaker = zip(*batch)
        out = [None] * 7
        if token[0] is not None:
            out[1] = torch.tensor(token_len, dtype=torch.long)
            longest = max(x.shape[-1] for x in token)
            out[0] = torch.stack([F.pad(x, (0, longest - x.shape[-1]), value=VQLatent.PAD) for x in token], dim=0)
        if spect[0] is not None:
            out[3] = torch.tensor(spect_len, dtype=torch.long)
            longest = max(x.shape[-1] for x in spect)
            out[2] = torch.stack([F.pad(x, (0, longest - x.shape[-1]), value=math.log(1e-7)) for x in spect], dim=0)
        if audio[0] is not None:
            out[5] = torch.tensor(audio_len, dtype=torch.long)
            longest = max(x.shape[-1] for x in audio)
            out[4] = torch.stack([F.pad(x, (0, longest - x.shape[-1])) for x in audio], dim=0).unsqueeze(1)
        if speaker[0] is not None:
            out[6] = torch.stack(speaker, dim=0)
        return tuple(out)
