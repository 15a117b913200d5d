"""Encode-only pass over a dataset with a trained VQ-VAE -> VQ-Latent files (reference scripts/generate_vq_dataset.py).

Run from the package root, like the reference:
    python -m scripts.generate_vq_dataset --log_dir ./logs/vqvae --ckpt_num 32500 --dump_dir ./data/VQ-Latent \
        --batch_size 8 --n_processes 8 --n_workers 4

Per utterance ``<split>/<index:05d>.pkl`` = ``{"x": [float...], "q": [int...]}`` (plain lists, generate_vq_dataset.py:84-91),
``metadata.json`` = ``{"compression_factor", "vocab_size"}`` (:215-220).  The encoder and the nearest-code search run in
libsmt_hip.so (``VQVAE.encode_and_quantize``); the code indices are the exact argmin, so the files are bit-reproducible.
Instead of the reference's matplotlib / librosa / soundfile artefacts (absent offline) the script writes
``<split>_histogram.json`` (code usage), ``sanity.wav`` (16-bit PCM via the stdlib) and ``sanity_mel.npz`` (log-mel of the
original and of the decoded clip, from the HIP mel kernel).
"""
import argparse
import json
import logging
import multiprocessing
import os
import random
import wave
from collections import Counter

import numpy as np
import torch

from datasets.vqlatent import dump_plain_pickle, load_plain_pickle
from utils import config as cfglib
from utils.commons import get_dataloaders, to_device

logger = logging.getLogger(__name__)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--log_dir", type=str, required=True, help="Log directory of training")
    p.add_argument("--ckpt_num", type=str, required=True, help="Checkpoint number to load (or 'last')")
    p.add_argument("--dump_dir", type=str, default="./data/VQ-Latent", help="Directory to dump VQ dataset")
    p.add_argument("--batch_size", type=int, default=8, help="Batch size for inference")
    p.add_argument("--n_processes", type=int, default=8, help="Number of processes to save pickle files with")
    p.add_argument("--n_workers", type=int, default=4, help="Number of dataloader workers")
    return p.parse_args(argv)


def dump_item_to_pickle(index, x, xl, q, ql, dump_dir):
    """One utterance: unpadded samples and codes as Python lists; returns the code histogram of the item."""
    x = x.reshape(-1)[:int(xl)].tolist()
    q = q.reshape(-1)[:int(ql)].tolist()
    dump_plain_pickle({"x": x, "q": q}, os.path.join(dump_dir, f"{index:05d}.pkl"))
    return Counter(q)


def generate_and_dump_dataset(dataloader, model, pool, dump_dir, split, device="cuda"):
    os.makedirs(os.path.join(dump_dir, split), exist_ok=True)
    batch_size = dataloader.batch_size
    counter = Counter()
    for i, batch in enumerate(dataloader):
        batch = to_device(batch, device)
        x, x_lengths = batch[4], batch[5]
        codes, code_lens = model.encode_and_quantize(x, x_lengths)
        x_cpu, xl, q, ql = x[:, 0].cpu(), x_lengths.cpu(), codes.cpu(), code_lens.cpu()
        n = x.shape[0]
        args = list(zip(range(i * batch_size, i * batch_size + n), x_cpu, xl, q, ql, [os.path.join(dump_dir, split)] * n))
        per_item = pool.starmap(dump_item_to_pickle, args) if pool is not None else [dump_item_to_pickle(*a) for a in args]
        counter += sum(per_item, Counter())
    with open(os.path.join(dump_dir, f"{split}_histogram.json"), "w", encoding="utf-8") as f:
        json.dump({str(k): counter[k] for k in sorted(counter)}, f)
    return counter


def write_wav(path, samples, sample_rate):
    pcm = (np.clip(np.asarray(samples, dtype=np.float64), -1.0, 1.0) * 32767.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(int(sample_rate))
        f.writeframes(pcm.tobytes())


def sanity_check(model, config, dump_dir, device):
    """Decode one stored item back to audio (generate_vq_dataset.py:179-212)."""
    name = random.sample(sorted(os.listdir(os.path.join(dump_dir, "train"))), 1)[0]
    data = load_plain_pickle(os.path.join(dump_dir, "train", name))
    q = torch.tensor(data["q"], dtype=torch.long, device=device).unsqueeze(0)
    q_lengths = torch.tensor((q.shape[-1],), dtype=torch.long, device=device)
    xh = model.dequantize_and_decode(q, q_lengths).reshape(-1).float().cpu()
    x = torch.tensor(data["x"], dtype=torch.float32)[:xh.numel()]
    write_wav(os.path.join(dump_dir, "sanity.wav"), xh.numpy(), config.dataset.sample_rate)
    try:
        from datasets.transforms import MelSpectrogram
        ds = config.dataset
        mel = MelSpectrogram(sample_rate=ds.sample_rate, n_fft=ds.n_fft, win_length=ds.win_length, hop_length=ds.hop_length,
                             n_mels=ds.n_mels, f_min=0.0, f_max=8000.0).to(device)
        np.savez(os.path.join(dump_dir, "sanity_mel.npz"), x=mel(x.clamp(-1, 1).to(device)).cpu().numpy(),
                 xh=mel(xh.clamp(-1, 1).to(device)).cpu().numpy())
    except Exception as e:  # the artefact is a convenience; the dataset itself is complete
        logger.warning("sanity mel not written: %s", e)
    return name


def write_metadata(config, dump_dir):
    metadata = {"compression_factor": int(np.prod(np.array(config.model.strides_t) ** np.array(config.model.downs_t))),
                "vocab_size": int(config.model.l_bins)}
    with open(os.path.join(dump_dir, "metadata.json"), "w", encoding="utf-8") as f:
        json.dump(metadata, f)
    return metadata


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("generate_vq_dataset runs the encoder and the VQ search on MI355X (libsmt_hip.so); no GPU is visible")
    device = torch.device("cuda")
    config = cfglib.load(os.path.join(args.log_dir, "config.yaml"))
    config.train.n_gpus = 1
    config.train.batch_size = args.batch_size
    config.train.num_workers = args.n_workers
    config.dataset.segment_length = -1
    from models.vqvae.vqvae import VQVAE
    ckpt = torch.load(os.path.join(args.log_dir, "ckpts", f"ckpt.{args.ckpt_num}.pt"), map_location=device, weights_only=True)
    model = VQVAE(config).to(device)
    model.load_state_dict(ckpt["model"])
    model.eval()
    config.dataset.use_spect = False
    config.dataset.use_token = False
    train_loader, val_loader = get_dataloaders(config, shuffle_train=False)
    os.makedirs(args.dump_dir, exist_ok=True)
    pool = multiprocessing.Pool(processes=args.n_processes) if args.n_processes > 1 else None
    try:
        generate_and_dump_dataset(train_loader, model, pool, args.dump_dir, "train", device)
        generate_and_dump_dataset(val_loader, model, pool, args.dump_dir, "val", device)
    finally:
        if pool is not None:
            pool.close(); pool.join()
    sanity_check(model, config, args.dump_dir, device)
    write_metadata(config, args.dump_dir)
    logger.info("Done")


if __name__ == "__main__":
    main()
