"""Helpers of the train loop (reference utils/train_utils.py): barrier, seeding, running
averages, scalar logging, checkpoints.  The val-time spectrogram / audio dumps of the
reference need librosa + soundfile + matplotlib and are outside the hot path."""
import json
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist

logger = logging.getLogger(__name__)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def seed_all_rng(seed, cuda=True):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if cuda and torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class ScalarWriter:
    """TensorBoard ``SummaryWriter`` when tensorboard is importable, JSON-lines otherwise
    (same ``add_scalar`` / ``close`` surface; train_utils.py:135-145)."""

    def __init__(self, log_dir):
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir)
        except Exception:  # tensorboard absent
            os.makedirs(log_dir, exist_ok=True)
            self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a", encoding="utf-8")

    def add_scalar(self, tag, value, step):
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)
        else:
            self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            self._f.flush()

    def close(self):
        (self._tb or self._f).close()


def accumulate_stats(over_n_steps, loss_dict, metrics_dict, accumulated_loss, accumulated_metrics):
    """Running means of every ``*loss*`` entry and every metric (train_utils.py:120-132).
    All scalars of a step are fetched with ONE device->host copy instead of one per key."""
    loss_keys = [k for k in loss_dict if "loss" in k]
    metric_keys = list(metrics_dict)
    if not loss_keys and not metric_keys:
        return
    stacked = torch.stack([loss_dict[k].detach().float().reshape(()) for k in loss_keys] +
                          [torch.as_tensor(metrics_dict[k]).detach().float().reshape(()).to(loss_dict[loss_keys[0]].device)
                           for k in metric_keys]).cpu().tolist()
    for k, v in zip(loss_keys, stacked[:len(loss_keys)]):
        accumulated_loss[k] += v / over_n_steps
    for k, v in zip(metric_keys, stacked[len(loss_keys):]):
        accumulated_metrics[k] += v / over_n_steps


def log_stats(step_or_epoch, writer, losses, metrics, prefix="train"):
    for key, value in losses.items():
        writer.add_scalar(f"loss/{prefix}_{key}", value, step_or_epoch)
    for key, value in metrics.items():
        writer.add_scalar(f"metrics/{prefix}_{key}", value, step_or_epoch)


def save_checkpoint(config, global_step, epoch, model, ema, optimizer, scheduler):
    """Same dictionary layout as the reference (train_utils.py:148-171); ``epoch=-1`` writes
    ``ckpt.last.pt``.  The config is stored as a plain dict so the file loads with
    ``torch.load(..., weights_only=True)``."""
    name = "last" if epoch == -1 else global_step
    path = os.path.join(config.train.log_dir, "ckpts", f"ckpt.{name}.pt")
    torch.save({
        "config": config.to_dict() if hasattr(config, "to_dict") else dict(config),
        "model": model.state_dict(),
        "optim": optimizer.state_dict(),
        "sched": scheduler.state_dict(),
        "ema": ema.state_dict(),
        "step": global_step,
        "epoch": config.train.total_epochs if epoch == -1 else epoch,
        # not in the reference's checkpoints (train_utils.py:148-171): the codebook's EMA accumulators and the dropout step
        # counter are plain attributes there and are lost on resume (bottleneck.py:20-24,179); loaders tolerate the key
        "extra": extra_train_state(model),
    }, path)
    return path


def extra_train_state(model):
    """Training state that lives outside ``state_dict()``: per codebook ``k_sum`` / ``k_elem`` / ``init``, and the
    dropout step counter."""
    out = {}
    bottleneck = getattr(model, "bottleneck", None)
    for i, blk in enumerate(getattr(bottleneck, "level_blocks", [])):
        if getattr(blk, "init", False) and blk.k_sum is not None:
            out[f"bottleneck.level_blocks.{i}"] = {"k_sum": blk.k_sum.detach().clone(), "k_elem": blk.k_elem.detach().clone(),
                                                   "threshold": float(blk.threshold)}
    if hasattr(model, "_drop_seed"):
        out["drop_seed"] = int(model._drop_seed)
    return out


def restore_extra_train_state(model, extra):
    """Counterpart of ``extra_train_state``.  Without the extra entry (a checkpoint written by the reference) a codebook
    that was trained (non-zero ``k``) is kept with ``restore_k()`` (bottleneck.py:48-58) instead of being re-drawn from
    the first batch, which is what the reference's resume path silently does (``init`` is False after loading)."""
    extra = extra or {}
    bottleneck = getattr(model, "bottleneck", None)
    for i, blk in enumerate(getattr(bottleneck, "level_blocks", [])):
        st = extra.get(f"bottleneck.level_blocks.{i}")
        if st is not None:
            blk.k_sum = st["k_sum"].to(blk.k.device, torch.float32).clone()
            blk.k_elem = st["k_elem"].to(blk.k.device, torch.float32).clone()
            blk.threshold = float(st.get("threshold", blk.threshold))
            blk.init = True
        elif bool(blk.k.abs().sum() > 0):
            blk.restore_k(threshold=blk.threshold)
    if "drop_seed" in extra and hasattr(model, "_drop_seed"):
        model._drop_seed = int(extra["drop_seed"])


def write_wav(path, samples, sample_rate):
    """16-bit mono PCM through the stdlib (the reference uses soundfile, train_utils.py:267-277)."""
    import wave
    pcm = (np.clip(np.asarray(samples, dtype=np.float64), -1.0, 1.0) * 32767.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(int(sample_rate))
        f.writeframes(pcm.tobytes())


def write_png_gray(path, image):
    """8-bit greyscale PNG with zlib + struct only (the reference draws its grids with matplotlib + PIL)."""
    import struct
    import zlib
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def spects_to_grid(ys, yhs, n=4):
    """Pairs of spectrograms [n_mels, frames] -> one uint8 image: row i = ground truth | prediction, low frequencies at
    the bottom, a common grey scale per pair (train_utils.py:174-195 without matplotlib)."""
    rows = []
    for i in range(min(n, len(ys))):
        a, b = np.asarray(ys[i], dtype=np.float32), np.asarray(yhs[i], dtype=np.float32)
        frames = max(a.shape[1], b.shape[1])
        lo, hi = min(a.min(), b.min()), max(a.max(), b.max())
        pair = np.zeros((a.shape[0], 2 * frames + 4), dtype=np.uint8)
        for off, m in ((0, a), (frames + 4, b)):
            pair[:, off:off + m.shape[1]] = np.clip(255.0 * (m[::-1] - lo) / max(hi - lo, 1e-12), 0, 255).astype(np.uint8)
        rows += [pair, np.zeros((4, pair.shape[1]), dtype=np.uint8)]
    width = max(r.shape[1] for r in rows)
    return np.concatenate([np.pad(r, ((0, 0), (0, width - r.shape[1]))) for r in rows], axis=0)


@torch.no_grad()
def save_audio_and_computed_spect(config, global_step, writer, audio, audio_pred, n=4):
    """Validation artefacts of a waveform model (reference train_utils.py:249-304): the first clip and its reconstruction as
    wav files, and a grid of the log-mel spectrograms of the first `n` pairs.  The mel front end is this build's HIP
    kernel (the reference's own MelSpectrogram, log of the mel-filtered MAGNITUDE) instead of librosa's power-dB
    spectrogram, which is not available offline; the image is a greyscale PNG."""
    from datasets.transforms import MelSpectrogram
    ds, log_dir = config.dataset, config.train.log_dir
    for sub in ("audio", "spect"):
        os.makedirs(os.path.join(log_dir, sub), exist_ok=True)
    audio, audio_pred = audio.detach().float().clamp(-1, 1), audio_pred.detach().float().clamp(-1, 1)
    write_wav(os.path.join(log_dir, "audio", f"val_audio_{global_step}_gt.wav"), audio[0].cpu().numpy(), ds.sample_rate)
    write_wav(os.path.join(log_dir, "audio", f"val_audio_{global_step}_pred.wav"), audio_pred[0].cpu().numpy(), ds.sample_rate)
    k = min(n, audio.shape[0])
    device = audio.device if audio.is_cuda else torch.device("cuda")
    mel = MelSpectrogram(sample_rate=ds.sample_rate, n_fft=ds.n_fft, win_length=ds.win_length, hop_length=ds.hop_length,
                         n_mels=ds.n_mels, f_min=0.0, f_max=8000.0).to(device)
    spect, spect_pred = mel(audio[:k].to(device)).cpu().numpy(), mel(audio_pred[:k].to(device)).cpu().numpy()
    grid = spects_to_grid(spect, spect_pred, n=k)
    path = os.path.join(log_dir, "spect", f"val_spect_{global_step}.png")
    write_png_gray(path, grid)
    tb = getattr(writer, "_tb", None)
    if tb is not None:
        tb.add_image("mel/val", grid, global_step, dataformats="HW")
        tb.add_audio("audio/val_gt", audio[0].cpu(), global_step=global_step, sample_rate=ds.sample_rate)
        tb.add_audio("audio/val_pred", audio_pred[0].cpu(), global_step=global_step, sample_rate=ds.sample_rate)
    return path


def print_top_level_summary(model):
    rows = []
    for name, module in model.named_children():
        n_params = sum(p.numel() for p in module.parameters() if p.requires_grad)
        n_buffers = sum(b.numel() for b in module.buffers())
        rows.append(f"  {name:<20s} {type(module).__name__:<32s} params {n_params:>12,d}  buffers {n_buffers:>10,d}")
    total = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("\n".join(rows) + f"\n  trainable parameters: {total:,d} ({total * 4e-6:,.1f} MB fp32)\n")
