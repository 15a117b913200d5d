"""Ancestral samples from a trained TransformerLM (reference scripts/sample_from_lm.py).

    python -m scripts.sample_from_lm --log_dir ./logs/transformer_lm --ckpt_num 5000 --dump_dir ./outputs \
        --n_samples 4 --n_steps 512 [--sigma 1.0]

Writes ``<dump_dir>/<ModelClass>@<ckpt>/sample_<i>.wav``, ``mel_spectrograms.png`` and ``tokens.txt`` like the reference.
The sampling loop, the dequantisation and the VQ-VAE decoder run on MI355X through libsmt_hip.so (`TransformerLM.sample`);
the spectrogram image is this build's log-mel (HIP kernel) as a greyscale PNG -- librosa / matplotlib / soundfile /
tabulate are not needed."""
import argparse
import logging
import os

import numpy as np
import torch

from utils import config as cfglib
from utils.commons import get_model
from utils.train_utils import write_png_gray, write_wav

logger = logging.getLogger(__name__)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--log_dir", type=str, required=True, help="Log directory of training")
    p.add_argument("--ckpt_num", type=int, required=True, help="Checkpoint number to load")
    p.add_argument("--dump_dir", type=str, default="./outputs", help="Directory to dump the samples")
    p.add_argument("--n_samples", type=int, default=4, help="Batch size for inference")
    p.add_argument("--n_steps", type=int, default=1024, help="Number of codes to sample")
    p.add_argument("--sigma", type=float, default=1.0, help="Sampling temperature")
    return p.parse_args(argv)


def mel_grid(spects):
    """[n_mels, frames] log-mels stacked top to bottom, low frequencies at the bottom of each, one grey scale per sample."""
    rows = []
    for m in spects:
        m = np.asarray(m, dtype=np.float32)[::-1]
        rows += [np.clip(255.0 * (m - m.min()) / max(float(m.max() - m.min()), 1e-12), 0, 255).astype(np.uint8),
                 np.zeros((4, m.shape[1]), dtype=np.uint8)]
    return np.concatenate(rows, axis=0)


def tokens_table(q):
    """Plain-text table: one row per sample, a header of step numbers (the reference formats it with tabulate)."""
    width = max(len(str(int(q.max()))), len(str(q.shape[1] - 1))) + 2
    lines = ["".join(f"{s:>{width}d}" for s in range(q.shape[1])), "".join("-" * (width - 2) + "  " for _ in range(q.shape[1]))]
    lines += ["".join(f"{int(v):>{width}d}" for v in row) for row in q.tolist()]
    return "\n".join(lines) + "\n"


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("sample_from_lm runs the language model and the VQ-VAE decoder on MI355X (libsmt_hip.so); no GPU is visible")
    device = torch.device("cuda")
    config = cfglib.load(os.path.join(args.log_dir, "config.yaml"))
    config.train.n_gpus = 1
    ckpt = torch.load(os.path.join(args.log_dir, "ckpts", f"ckpt.{args.ckpt_num}.pt"), map_location=device, weights_only=True)
    model, _ = get_model(config, device=device)
    model.load_state_dict(ckpt["model"])
    model.eval()
    dump_dir = os.path.join(args.dump_dir, f"{type(model).__name__}@{args.ckpt_num}")
    os.makedirs(dump_dir, exist_ok=True)

    x_samples, q_samples = model.sample(batch_size=args.n_samples, n_steps=args.n_steps, device=device, sigma=args.sigma)
    logger.info("Generated token samples")

    from datasets.transforms import MelSpectrogram
    ds = config.dataset
    mel = MelSpectrogram(sample_rate=ds.sample_rate, n_fft=ds.n_fft, win_length=ds.win_length, hop_length=ds.hop_length,
                         n_mels=ds.n_mels, f_min=0.0, f_max=8000.0).to(device)
    audio = x_samples.float().clamp(-1, 1)
    for i in range(args.n_samples):
        write_wav(os.path.join(dump_dir, f"sample_{i}.wav"), audio[i].cpu().numpy(), ds.sample_rate)
    write_png_gray(os.path.join(dump_dir, "mel_spectrograms.png"), mel_grid(mel(audio).cpu().numpy()))
    with open(os.path.join(dump_dir, "tokens.txt"), "w", encoding="utf-8") as f:
        f.write(tokens_table(q_samples.cpu()))
    logger.info("Saved audio, spectrograms and tokens under %s", dump_dir)
    return dump_dir


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main()
