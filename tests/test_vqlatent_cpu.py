"""VQ-Latent on-disk format and dataset (reference scripts/generate_vq_dataset.py:84-91,215-220 and
datasets/vqlatent.py:61-142) -- host logic only, no GPU."""
import collections
import json
import os
import pickle

import pytest
import torch


def _cfg(path, segment_length=-1, remove_consecutive=False, vocab=64):
    from utils import config as C
    return C.create({"dataset": {"dataset_path": str(path), "segment_length": segment_length,
                                 "remove_consecutive": remove_consecutive, "use_token": True, "use_spect": False,
                                 "use_audio": True},
                     "model": {"vocab_size": vocab}})


def _make(tmp_path, items, cf=4, vocab=64):
    from scripts.generate_vq_dataset import dump_item_to_pickle
    os.makedirs(tmp_path / "train")
    hist = collections.Counter()
    for i, (x, q) in enumerate(items):
        pad_x = torch.cat([torch.tensor(x), torch.zeros(5)])          # batch padding must not reach the file
        pad_q = torch.cat([torch.tensor(q), torch.full((3,), 63)])
        hist += dump_item_to_pickle(i, pad_x, torch.tensor(len(x)), pad_q, torch.tensor(len(q)), str(tmp_path / "train"))
    with open(tmp_path / "metadata.json", "w") as f:
        json.dump({"compression_factor": cf, "vocab_size": vocab}, f)
    return hist


def test_file_format_is_plain_lists_and_loads_without_executing_anything(tmp_path):
    from datasets.vqlatent import load_plain_pickle
    items = [([0.25, -0.5, 0.125, 1.0] * 2, [3, 3]), ([0.0] * 12, [7, 0, 9])]
    hist = _make(tmp_path, items)
    assert hist == collections.Counter({3: 2, 7: 1, 0: 1, 9: 1})
    assert sorted(os.listdir(tmp_path / "train")) == ["00000.pkl", "00001.pkl"]
    for i, (x, q) in enumerate(items):
        with open(tmp_path / "train" / f"{i:05d}.pkl", "rb") as f:
            raw = pickle.load(f)                                       # our own file: exactly what the reference reads
        assert raw == {"x": x, "q": q} and type(raw["x"]) is list and type(raw["q"][0]) is int
        assert load_plain_pickle(tmp_path / "train" / f"{i:05d}.pkl") == raw
    evil = tmp_path / "evil.pkl"
    with open(evil, "wb") as f:
        pickle.dump({"x": collections.Counter([1])}, f)              # needs a GLOBAL opcode
    with pytest.raises(pickle.UnpicklingError):
        load_plain_pickle(evil)


def test_dataset_items_offsets_and_collate(tmp_path):
    from datasets.vqlatent import VQLatent
    cf = 4
    items = [(list(range(8)), [5, 6]), (list(range(100, 112)), [1, 1, 2])]
    _make(tmp_path, [([float(v) for v in x], q) for x, q in items], cf=cf)
    ds = VQLatent(_cfg(tmp_path), "train")
    assert len(ds) == 2 and ds.metadata == {"compression_factor": cf, "vocab_size": 64}
    tok, tok_len, spect, spect_len, audio, audio_len, speaker = ds[1]
    assert tok.tolist() == [VQLatent.BOS, 1 + 2, 1 + 2, 2 + 2] and tok_len == 4          # <bos> + codes shifted by OFFSET
    assert audio.tolist() == [float(v) for v in range(100, 112)] and audio_len == 12 and spect is None and speaker is None
    batch = VQLatent.collate([ds[0], ds[1]])
    assert batch[0].tolist() == [[1, 7, 8, VQLatent.PAD], [1, 3, 3, 4]] and batch[1].tolist() == [3, 4]
    assert batch[4].shape == (2, 1, 12) and batch[5].tolist() == [8, 12] and batch[4][0, 0, 8:].abs().sum() == 0
    assert batch[2] is None and batch[6] is None
    # consecutive duplicates removed before the offset
    ds2 = VQLatent(_cfg(tmp_path, remove_consecutive=True), "train")
    assert ds2[1][0].tolist() == [1, 3, 4]
    with pytest.raises(AssertionError):
        VQLatent(_cfg(tmp_path, vocab=128), "train")


def test_segment_crop_keeps_audio_aligned_and_pads_short_items(tmp_path):
    import random
    from datasets.vqlatent import VQLatent
    cf = 4
    q_long = list(range(10, 30))                                     # 20 codes, audio sample i belongs to code i // cf
    x_long = [float(i // cf) for i in range(20 * cf)]
    _make(tmp_path, [(x_long, q_long), ([1.0] * 8, [40, 41])], cf=cf)
    ds = VQLatent(_cfg(tmp_path, segment_length=6), "train")
    random.seed(3)
    tok, tok_len, _, _, audio, audio_len, _ = ds[0]
    assert tok_len == 7 and tok.shape[-1] == 8 and tok[0] == VQLatent.BOS and tok[-1] == VQLatent.PAD
    start = int(tok[1]) - VQLatent.OFFSET - 10
    assert tok[1:7].tolist() == [c + VQLatent.OFFSET for c in q_long[start:start + 6]]
    assert audio_len == 6 * cf and audio.tolist() == x_long[start * cf:(start + 6) * cf]
    tok, tok_len, _, _, audio, audio_len, _ = ds[1]                  # shorter than the segment: padded
    assert tok.tolist() == [1, 42, 43, 0, 0, 0, 0, 0] and tok_len == 3 and audio.shape[-1] == 6 * cf and audio_len == 8


def test_metadata_matches_the_reference_keys(tmp_path):
    from scripts.generate_vq_dataset import write_metadata
    from utils import config as C
    cfg = C.create({"model": {"strides_t": [2, 2, 2], "downs_t": [3, 2, 2], "l_bins": 1024}})
    assert write_metadata(cfg, str(tmp_path)) == {"compression_factor": 128, "vocab_size": 1024}
    assert json.load(open(tmp_path / "metadata.json")) == {"compression_factor": 128, "vocab_size": 1024}
