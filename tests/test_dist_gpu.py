"""The data-parallel train step on the GPU: two ranks (both on cuda:0, gloo carrying the device tensors
because one card cannot host two RCCL ranks) run the real model through libsmt_hip.so with GradSync
and the one-buffer codebook all-reduce.  After every step all replicas must be bit-identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd")
REPO = os.path.dirname(PKG)


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, PKG); sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import train as trainlib
    from oracle import vqvae_oracle as orc
    from smt_amd.dist import GradSync
    from utils import config as C
    from utils.commons import get_model, get_optimizer
    from utils.train_utils import seed_all_rng
    cfg = C.merge(C.load(os.path.join(PKG, "configs/models/vqvae.yaml")),
                  C.load(os.path.join(PKG, "configs/datasets/synthetic_ljspeech.yaml")),
                  C.create({"train": {"batch_size": 2, "n_gpus": world, "ema": False, "grad_clip_norm": None, "seed": 0}}))
    cfg.model.update(C.create(dict(width=16, emb_width=32, l_bins=64, multipliers=[1, 1, 1])))
    cfg.model.loss.linf_topk = 128
    seed_all_rng(100 + rank)                      # DIFFERENT init per rank: get_model must broadcast rank 0's
    model, ema = get_model(cfg, torch.device("cuda", 0), rank)
    opt, sched = get_optimizer(cfg, model)
    sync = GradSync(model.parameters(), bucket_bytes=64 << 10)
    assert len(sync.buckets) > 2
    model.train()
    losses = []
    for step in range(3):
        x = orc.synthetic_clip_batch(2, 8192, 1000 * rank + step).cuda()      # each rank its own utterances
        lens = torch.tensor([8192, 4096 + 2048 * rank]).cuda()
        loss_dict, metrics = trainlib.train_step(global_step=step, batch=[None, None, None, None, x, lens, None],
                                                 config=cfg, model=model, ema=ema, optimizer=opt, scheduler=sched,
                                                 device=torch.device("cuda", 0), rank=rank, grad_sync=sync)
        losses.append(loss_dict["loss"].item())
    blk = model.bottleneck.level_blocks[0]
    torch.save({"params": torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()]),
                "k": blk.k.cpu(), "k_sum": blk.k_sum.cpu(), "k_elem": blk.k_elem.cpu(), "losses": torch.tensor(losses),
                "grads": sync.flat.cpu().clone()}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_train_steps_keep_replicas_identical(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    for name in ("params", "k", "k_sum", "k_elem", "grads"):
        assert torch.equal(r0[name], r1[name]), name                # replicas bit-identical after 3 steps
    assert torch.isfinite(r0["params"]).all() and not torch.equal(r0["losses"], r1["losses"])   # different data per rank


def _rccl_worker(rank, world, port, tmp):
    """One rank, backend nccl (= RCCL): the collectives really go through the library the multi-GPU bench uses."""
    import sys
    sys.path.insert(0, PKG); sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    import train as trainlib
    from oracle import vqvae_oracle as orc
    from smt_amd.dist import GradSync
    from utils import config as C
    from utils.commons import get_model, get_optimizer
    from utils.train_utils import seed_all_rng

    def run(distributed):
        if distributed:
            dist.init_process_group("nccl", rank=0, world_size=1)
        cfg = C.merge(C.load(os.path.join(PKG, "configs/models/vqvae.yaml")),
                      C.load(os.path.join(PKG, "configs/datasets/synthetic_ljspeech.yaml")),
                      C.create({"train": {"batch_size": 2, "n_gpus": 1, "ema": False, "grad_clip_norm": None, "seed": 0}}))
        cfg.model.update(C.create(dict(width=16, emb_width=32, l_bins=64, multipliers=[1, 1, 1])))
        cfg.model.loss.linf_topk = 128
        seed_all_rng(7)
        model, ema = get_model(cfg, torch.device("cuda", 0), 0)
        opt, sched = get_optimizer(cfg, model)
        # always_reduce: hooks + asynchronous bucket all-reduces are issued on the nccl (= RCCL) backend even though the
        # world has one rank, so the stream-ordered wait() path the multi-GPU bench relies on is the one exercised
        sync = GradSync(model.parameters(), bucket_bytes=64 << 10, always_reduce=True, timing=True) if distributed else None
        model.train()
        losses = []
        for step in range(3):
            x = orc.synthetic_clip_batch(2, 8192, step).cuda()
            lens = torch.tensor([8192, 6144]).cuda()
            loss_dict, _ = trainlib.train_step(global_step=step, batch=[None, None, None, None, x, lens, None], config=cfg,
                                               model=model, ema=ema, optimizer=opt, scheduler=sched,
                                               device=torch.device("cuda", 0), rank=0, grad_sync=sync)
            losses.append(loss_dict["loss"].item())
        params = torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()])
        if distributed:
            assert sync.reduce and len(sync.buckets) > 2 and sync.exposed_ms() is not None
            dist.destroy_process_group()
        return torch.tensor(losses), params

    l1, p1 = run(True)
    l0, p0 = run(False)
    torch.save({"l0": l0, "l1": l1, "p0": p0, "p1": p1}, os.path.join(tmp, "rccl.pt"))


def test_single_rank_rccl_group_matches_the_plain_step(tmp_path):
    """The RCCL code path (GradSync buckets, codebook one-buffer all-reduce, broadcast) on the one GPU a test box
    has: a world of one rank must reproduce the non-distributed train steps exactly."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl.pt", weights_only=True)
    assert torch.isfinite(r["p1"]).all()
    assert torch.allclose(r["l0"], r["l1"], rtol=1e-5, atol=1e-6), (r["l0"], r["l1"])
    # f32 atomics (codebook sums, STFT overlap-add) make two runs differ in the last bits, and AdamW (eps = 1e-9) turns
    # a last-bit difference of a near-zero gradient into a step of the order of the learning rate
    diff = (r["p0"] - r["p1"]).abs()
    assert float(diff.mean()) <= 1e-5 and float(diff.max()) <= 5e-3
