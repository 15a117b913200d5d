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


def print_top_level_summary(model):
    rows = []
    for name, module in model.named_children():
        n_params = sum(p.numel() for p in module.parameters() if p.requires_grad)
        n_buffers = sum(b.numel() for b in module.buffers())
        rows.append(f"  {name:<20s} {type(module).__name__:<32s} params {n_params:>12,d}  buffers {n_buffers:>10,d}")
    total = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("\n".join(rows) + f"\n  trainable parameters: {total:,d} ({total * 4e-6:,.1f} MB fp32)\n")
