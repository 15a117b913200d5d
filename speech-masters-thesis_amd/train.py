"""Train driver, flag-compatible with the reference's train.py (train.py:47-79).

Run from this directory (configs/ is resolved relative to the working directory, like the
reference):   python train.py --model vqvae --dataset synthetic_ljspeech --batch_size 4

One process per GPU.  Multi-GPU: either launch under torchrun
(``python -m torch.distributed.run --nproc-per-node N train.py ...``) or pass ``--n_gpus N`` and
let this script spawn the ranks (the reference's ``mp.spawn`` behaviour, train.py:568).
"""
import argparse
import logging
import os
import sys
from collections import defaultdict

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # ahead of any pip package named `datasets`

from utils import config as cfglib  # noqa: E402
from utils.commons import get_dataloaders, get_model, get_optimizer, setup_logdir, to_device  # noqa: E402
from utils.train_utils import (ScalarWriter, accumulate_stats, barrier, log_stats,  # noqa: E402
                               print_top_level_summary, restore_extra_train_state, save_audio_and_computed_spect,
                               save_checkpoint, seed_all_rng)

logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s: %(message)s")
logger = logging.getLogger("train")


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model", type=str, default="vqvae", help="Name of model config in configs/models")
    p.add_argument("--dataset", type=str, default="ljspeech", help="Name of dataset config in configs/datasets")
    p.add_argument("--log_dir", type=str, default="./logs/vqvae")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--ema", default=False, action="store_true")
    p.add_argument("--grad_clip_norm", type=float, default=None)
    p.add_argument("--fp16", default=False, action="store_true",
                   help="accepted for compatibility; reduced precision is selected by model.compute_dtype (bf16)")
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--n_gpus", type=int, default=-1)
    p.add_argument("--total_epochs", type=int, default=1000)
    p.add_argument("--load_ckpt", type=str, default=None)
    p.add_argument("--ckpt_every_n_steps", type=int, default=10000)
    p.add_argument("--log_every_n_steps", type=int, default=10)
    p.add_argument("--eval_every_n_epochs", type=int, default=5)
    p.add_argument("--run_sanity_val_epoch", default=False, action="store_true")
    return p.parse_args(argv)


def train_step(*, global_step, batch, config, model, ema, optimizer, scheduler, device, rank=0, grad_sync=None):
    """zero_grad -> supervised_step -> NaN guard -> backward -> [grad all-reduce] -> [clip] -> optimiser
    -> scheduler -> parameter EMA (reference train.py:82-143, full-precision branch)."""
    batch = to_device(batch, device)
    if grad_sync is not None:
        grad_sync.zero_grad()
    else:
        optimizer.zero_grad()
    loss_dict, metrics_dict = model.supervised_step(batch)
    loss = loss_dict["loss"]
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish()
    if config.train.grad_clip_norm:
        torch.nn.utils.clip_grad_norm_(model.parameters(), config.train.grad_clip_norm)
    # NaN guard: the reference tests the loss before backward (train.py:124); testing after the
    # collectives are queued keeps the device->host sync off the critical path, same abort semantics.
    if torch.isnan(loss):
        print(dict(**{k: v for k, v in loss_dict.items() if k.startswith("loss")}, **metrics_dict,
                   STEP=global_step, RANK=rank))
        raise RuntimeError(f"Nan detected in loss at step {global_step}")
    optimizer.step()
    scheduler.step()
    ema.step()
    return loss_dict, metrics_dict


def train_epoch(*, global_step, epoch, config, model, ema, optimizer, scheduler, train_dataloader, writer, device,
                rank=0, grad_sync=None):
    losses, metrics = defaultdict(float), defaultdict(float)
    model.train()
    for batch in train_dataloader:
        loss_dict, metrics_dict = train_step(global_step=global_step, batch=batch, config=config, model=model,
                                             ema=ema, optimizer=optimizer, scheduler=scheduler, device=device,
                                             rank=rank, grad_sync=grad_sync)
        global_step += 1
        if rank == 0:
            accumulate_stats(config.train.log_every_n_steps, loss_dict, metrics_dict, losses, metrics)
            if global_step % config.train.log_every_n_steps == 0:
                log_stats(global_step, writer, losses, metrics)
                logger.info("step %d %s", global_step,
                            " ".join(f"{k}={v:.4f}" for k, v in {**losses, **metrics}.items()))
                losses, metrics = defaultdict(float), defaultdict(float)
            if global_step % config.train.ckpt_every_n_steps == 0:
                save_checkpoint(config, global_step, epoch, model, ema, optimizer, scheduler)
    return global_step, epoch + 1


@torch.no_grad()
def val_epoch(*, epoch, config, model, ema, val_dataloader, writer, device):
    losses, metrics = defaultdict(float), defaultdict(float)
    model.eval()
    ema.swap()
    ys, yhs = [], []
    for batch in val_dataloader:
        loss_dict, metrics_dict = model.supervised_step(to_device(batch, device))
        accumulate_stats(len(val_dataloader), loss_dict, metrics_dict, losses, metrics)
        if "y" in loss_dict and "yh" in loss_dict and len(ys) < 4:          # train.py:274-275
            ys += list(loss_dict["y"][:4 - len(ys)]); yhs += list(loss_dict["yh"][:4 - len(yhs)])
    ema.swap()
    log_stats(epoch, writer, losses, metrics, prefix="val")
    from models.base import TokenToWaveformModel, WaveformReconstructionModel
    if isinstance(model, (TokenToWaveformModel, WaveformReconstructionModel)) and ys:     # train.py:296-299
        n = min(len(ys), len(yhs))
        t = min(min(v.shape[-1] for v in ys[:n]), min(v.shape[-1] for v in yhs[:n]))
        save_audio_and_computed_spect(config, epoch, writer, torch.stack([v[:t] for v in ys[:n]]),
                                      torch.stack([v[:t] for v in yhs[:n]]), n=n)
    return {**losses, **metrics}


def train(*, global_step, epoch, config, model, ema, optimizer, scheduler, train_dataloader, val_dataloader,
          writer, device, rank=0, grad_sync=None):
    assert callable(getattr(model, "supervised_step", None)), \
        f"Model type {type(model).__name__} doesn't have forward handle `supervised_step`"
    barrier()
    if rank == 0:
        print_top_level_summary(model)
    if config.train.run_sanity_val_epoch and rank == 0:
        logger.info("Sanity val epoch done: %s", val_epoch(epoch=epoch, config=config, model=model, ema=ema,
                                                           val_dataloader=val_dataloader, writer=writer,
                                                           device=device))
    while epoch < config.train.total_epochs:
        global_step, epoch = train_epoch(global_step=global_step, epoch=epoch, config=config, model=model, ema=ema,
                                         optimizer=optimizer, scheduler=scheduler,
                                         train_dataloader=train_dataloader, writer=writer, device=device, rank=rank,
                                         grad_sync=grad_sync)
        if epoch % config.train.eval_every_n_epochs == 0 and rank == 0:
            logger.info("epoch %d val: %s", epoch, val_epoch(epoch=epoch, config=config, model=model, ema=ema,
                                                             val_dataloader=val_dataloader, writer=writer,
                                                             device=device))
        barrier()
    if rank == 0:
        save_checkpoint(config, global_step, -1, model, ema, optimizer, scheduler)
        writer.close()


def load_checkpoint(path, model, optimizer, scheduler, ema, device):
    ckpt = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ckpt["model"])
    restore_extra_train_state(model, ckpt.get("extra"))     # codebook accumulators, dropout counter (or restore_k)
    from smt_amd import convops
    convops.invalidate_packed_weights()
    optimizer.load_state_dict(ckpt["optim"])
    scheduler.load_state_dict(ckpt["sched"])
    ema.load_state_dict(ckpt["ema"])
    return ckpt["step"], ckpt["epoch"]


def run_rank(rank, world_size, config, spawned=False):
    """Body of one process == one GPU (reference train_multi / train_single, train.py:389-508)."""
    cuda = torch.cuda.is_available()
    seed_all_rng(config.train.seed, cuda=cuda)
    if world_size > 1:
        if spawned:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "12355")
        local = int(os.environ.get("LOCAL_RANK", rank))
        if cuda:
            torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl" if cuda else "gloo", init_method="env://", rank=rank,
                                world_size=world_size)
        device = torch.device("cuda", local) if cuda else torch.device("cpu")
    else:
        device = torch.device("cuda") if cuda else torch.device("cpu")
    if device.type != "cuda":
        raise RuntimeError("the VQ-VAE hot path runs on MI355X only (libsmt_hip.so); no GPU is visible")

    writer = ScalarWriter(config.train.log_dir) if rank == 0 else None
    model, ema = get_model(config, device, rank)
    optimizer, scheduler = get_optimizer(config, model)
    train_loader, val_loader = get_dataloaders(config, rank, world_size)
    global_step = epoch = 0
    if config.train.load_ckpt:
        global_step, epoch = load_checkpoint(config.train.load_ckpt, model, optimizer, scheduler, ema, device)
    grad_sync = None
    if world_size > 1:
        from smt_amd.dist import GradSync
        grad_sync = GradSync(model.parameters())
    logger.info("[rank %d / %d] initialised on %s", rank, world_size, device)
    try:
        train(global_step=global_step, epoch=epoch, config=config, model=model, ema=ema, optimizer=optimizer,
              scheduler=scheduler, train_dataloader=train_loader, val_dataloader=val_loader, writer=writer,
              device=device, rank=rank, grad_sync=grad_sync)
    except KeyboardInterrupt:
        pass
    if world_size > 1:
        dist.destroy_process_group()


def build_config(args):
    model_config = cfglib.load(f"configs/models/{args.model}.yaml")
    dataset_config = cfglib.load(f"configs/datasets/{args.dataset}.yaml")
    train_config = cfglib.create({"train": {k: getattr(args, k) for k in (
        "log_dir", "seed", "batch_size", "ema", "grad_clip_norm", "fp16", "num_workers", "n_gpus", "total_epochs",
        "load_ckpt", "ckpt_every_n_steps", "log_every_n_steps", "eval_every_n_epochs", "run_sanity_val_epoch")}})
    return cfglib.merge(model_config, dataset_config, train_config)


def main(argv=None):
    args = parse_args(argv)
    config = build_config(args)
    max_gpus = torch.cuda.device_count()
    if config.train.n_gpus == -1:
        config.train.n_gpus = max_gpus
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if under_torchrun:
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        config.train.n_gpus = world
        if rank == 0:
            setup_logdir(config)
        run_rank(rank, world, config)
        return
    n_gpus = min(config.train.n_gpus, max_gpus)
    setup_logdir(config)
    if n_gpus <= 1:
        run_rank(0, 1, config)
    else:
        import torch.multiprocessing as mp
        mp.spawn(run_rank, args=(n_gpus, config, True), nprocs=n_gpus, join=True)


if __name__ == "__main__":
    main()
