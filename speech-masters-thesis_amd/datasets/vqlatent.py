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
        item = load_plain_pickle(os.path.join(self.dataset_path, self.split, self.pkl_files[index]))
        samples, codes = item["x"], item["q"]
        speaker = torch.tensor((item["speaker"],), dtype=torch.long) if "speaker" in item else None
        if self.remove_consecutive:
            codes = [run[0] for run in groupby(codes)]
        ratio, seg = self.metadata["compression_factor"], self.segment_length      # samples per code, codes per segment
        if 0 < seg < len(codes):
            first = random.randint(0, len(codes) - seg)                           # one window over codes and samples alike
            codes, samples = codes[first:first + seg], samples[first * ratio:(first + seg) * ratio]
        # <bos> in front; every code moves up by OFFSET so that 0 / 1 stay <pad> / <bos>
        token = torch.tensor([VQLatent.BOS] + [q + VQLatent.OFFSET for q in codes], dtype=torch.long)
        audio = torch.tensor(samples, dtype=torch.float32).reshape(-1)
        token_len, audio_len = token.numel(), audio.numel()
        if seg > 0:                                                               # short items fill up the segment (vqlatent.py:95-98)
            token = F.pad(token, (0, seg + 2 - token_len), value=VQLatent.PAD)
            audio = F.pad(audio, (0, seg * ratio - audio_len))
        if self.use_spect:
            raise NotImplementedError("use_spect: log-mel is computed on the device by datasets.transforms."
                                      "MelSpectrogram, not inside DataLoader workers")
        if not self.use_audio:
            audio = audio_len = None
        if not self.use_token:
            token = token_len = None
        return token, token_len, None, None, audio, audio_len, speaker

    @staticmethod
    def collate(batch):
        """Slots the config does not need stay None (vqlatent.py:114-142); the others are right-padded to the longest item."""
        def pad_stack(items, fill):
            width = max(t.shape[-1] for t in items)
            return torch.stack([F.pad(t, (0, width - t.shape[-1]), value=fill) for t in items], dim=0)

        token, token_len, spect, spect_len, audio, audio_len, speaker = zip(*batch)
        out = [None] * 7
        for slot, items, lens, fill in ((0, token, token_len, VQLatent.PAD), (2, spect, spect_len, math.log(1e-7)),
                                        (4, audio, audio_len, 0.0)):
            if items[0] is not None:
                out[slot] = pad_stack(items, fill)
                out[slot + 1] = torch.tensor(lens, dtype=torch.long)
        if out[4] is not None:
            out[4] = out[4].unsqueeze(1)
        if speaker[0] is not None:
            out[6] = torch.stack(speaker, dim=0)
        return tuple(out)
