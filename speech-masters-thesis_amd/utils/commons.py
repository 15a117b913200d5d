"""Factories behind train.py (reference utils/commons.py): the ``_import_`` registry for
models and datasets, optimiser / scheduler construction, log-dir set-up.

Differences from the reference, all deliberate:
  * data parallelism is explicit (``smt_amd.dist.GradSync``): the reference wraps the model
    in DDP but then calls ``model.module.supervised_step`` so gradients are never reduced
    (SURVEY.md D5); here gradients are mean-all-reduced over RCCL in buckets;
  * configs are ``utils.config.Config`` trees instead of OmegaConf.
"""
import importlib
import logging
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from utils import config as cfglib

logger = logging.getLogger(__name__)


def to_device(batch, device):
    return [b.to(device, non_blocking=True) if isinstance(b, torch.Tensor) else b for b in batch]


def _resolve(dotted):
    module, name = dotted.rsplit(".", 1)
    return getattr(importlib.import_module(module), name)


def get_model(config, device="cuda:0", rank=0):
    """(model, ema).  Also flips the dataset flags the model family does not need
    (reference commons.py:38-43)."""
    from models.base import (SpectrogramReconstructionModel, TokenToSpectrogramModel, TokenToWaveformModel,
                             WaveformReconstructionModel)
    from models.ema import EMA, DummyEMA
    model = _resolve(config.model["_import_"])(config).to(device)
    if isinstance(model, (TokenToWaveformModel, WaveformReconstructionModel)):
        config.dataset.use_spect = False
    if isinstance(model, (TokenToSpectrogramModel, SpectrogramReconstructionModel)):
        config.dataset.use_audio = False
    if isinstance(model, (WaveformReconstructionModel, SpectrogramReconstructionModel)):
        config.dataset.use_token = False
    if dist.is_initialized():
        from smt_amd.dist import broadcast_module
        broadcast_module(model, src=0)  # every rank starts from rank 0's weights and buffers
    if config.train.get("ema", False):
        n_gpus = max(int(config.train.get("n_gpus", 1)), 1)
        ema = EMA(model, mu=1 - (config.train.batch_size * n_gpus / 1000.0))
    else:
        ema = DummyEMA()
    return model, ema


def get_dataloaders(config, rank=0, world_size=1, shuffle_train=True):
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    dataset = _resolve(config.dataset["_import_"])
    train_set = dataset(config, split="train")
    common = dict(batch_size=config.train.batch_size, num_workers=config.train.num_workers, pin_memory=True,
                  drop_last=False, collate_fn=dataset.collate)
    if dist.is_initialized():
        sampler = DistributedSampler(train_set, num_replicas=world_size, rank=rank, shuffle=True)
        train_loader = DataLoader(train_set, sampler=sampler, **common)
    else:
        train_loader = DataLoader(train_set, shuffle=shuffle_train, **common)
    val_loader = DataLoader(dataset(config, split="val"), **common) if rank == 0 else None
    return train_loader, val_loader


def get_optimizer(config, model):
    opt = config.optimizer
    params = [p for p in model.parameters() if p.requires_grad]
    if opt.name == "adam":
        optimizer = torch.optim.AdamW(params, lr=float(opt.lr), betas=tuple(float(b) for b in opt.betas),
                                      weight_decay=float(opt.weight_decay), eps=float(opt.eps),
                                      fused=params[0].is_cuda)
    elif opt.name == "sgd":
        optimizer = torch.optim.SGD(params, lr=float(opt.lr), momentum=float(opt.momentum),
                                    weight_decay=float(opt.weight_decay))
    else:
        raise ValueError(f"Didn't recognize optimizer name {opt.name}")
    # (the packed operand copies of the conv weights follow every update by themselves: smt_amd.convops.training_forward)
    sched = config.get("scheduler", None)
    from utils import lr_scheduler as S
    if not sched:
        scheduler = S.DummyLR(optimizer)
    elif sched.name == "noam":
        # the reference reads config.model.d_model (utils/commons.py:151-155), which configs/models/glow_tts.yaml does not
        # have: the Noam scale of a GlowTTS run is its encoder width
        dim = config.model.get("d_model", None) or config.model.encoder.hidden_channels
        scheduler = S.NoamLR(optimizer, dim_model=dim, warmup_steps=sched.warmup_steps)
    elif sched.name == "linear":
        scheduler = S.LinearWarmupLR(optimizer, warmup_steps=sched.warmup_steps)
    elif sched.name == "cosine":
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config.train.total_steps)
    else:
        raise ValueError(f"Didn't recognize scheduler name {sched.name}")
    return optimizer, scheduler


def setup_logdir(config):
    log_dir = config.train.log_dir
    for sub in ("", "ckpts", "spect", "audio"):
        os.makedirs(os.path.join(log_dir, sub), exist_ok=True)
    cfglib.save(config, os.path.join(log_dir, "config.yaml"))
    logger.info("Set up logdir at %s", log_dir)
