"""Data-parallel path on CPU with the gloo backend (world_size 2): gradient mean all-reduce in
buckets, and the codebook-statistics choreography (ONE all-reduce carrying sums, counts and rank 0's
revival rows) against the multi-GPU oracle of SURVEY.md 8(e)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vqvae_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    port = _free_port()
    mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)          # same seed on every rank, like train.py


def _grad_sync_worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd"))
    from smt_amd.dist import GradSync, broadcast_module
    _init(rank, world, port)
    model = torch.nn.Sequential(torch.nn.Linear(7, 9), torch.nn.Tanh(), torch.nn.Linear(9, 5), torch.nn.Linear(5, 3))
    if rank == 1:                      # ranks start different; broadcast must fix that
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    broadcast_module(model, src=0)
    import copy
    twin = copy.deepcopy(model)                                # un-synced twin: this rank's LOCAL gradients
    sync = GradSync(model.parameters(), bucket_bytes=200)      # several tiny buckets
    assert len(sync.buckets) > 1
    g = torch.Generator().manual_seed(100 + rank)
    results = {}
    for step in range(2):
        x = torch.randn(4, 7, generator=g)
        sync.zero_grad()
        loss = model(x).pow(2).sum() * (rank + 1)
        loss.backward()            # bucket all-reduces are launched from the autograd hooks during this call
        twin.zero_grad()
        (twin(x).pow(2).sum() * (rank + 1)).backward()
        local = torch.cat([p.grad.detach().clone().reshape(-1) for p in twin.parameters()])
        sync.finish()
        synced = torch.cat([p.grad.detach().reshape(-1) for p in model.parameters()])
        results[f"local{step}"] = local
        results[f"synced{step}"] = synced.clone()
        assert all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in model.parameters())  # views of the flat buffer
    results["w"] = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.save(results, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_grad_sync_mean_allreduce_two_ranks(tmp_path):
    _spawn(_grad_sync_worker, 2, str(tmp_path))
    r0 = torch.load(tmp_path / "r0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=True)
    assert torch.equal(r0["w"], r1["w"])                         # broadcast made the replicas identical
    for step in range(2):
        mean = 0.5 * (r0[f"local{step}"] + r1[f"local{step}"])   # oracle: (1/W) sum_r grad_r  
        assert torch.allclose(r0[f"synced{step}"], mean, atol=1e-6)
        assert torch.equal(r0[f"synced{step}"], r1[f"synced{step}"])
        assert not torch.allclose(r0[f"local{step}"], r1[f"local{step}"])


def _cpu_vq_ops():
    """CPU stand-ins for the two HIP EMA kernels so the collective choreography of
    BottleneckBlock.update_k can run under gloo (the kernels themselves are GPU-tested)."""
    def ema_accumulate(x, idx, row_mask, k_bins, stats):
        d = x.shape[1]
        stats.zero_()
        sel = row_mask != 0 if row_mask is not None else torch.ones(x.shape[0], dtype=torch.bool)
        stats[:k_bins * d].view(k_bins, d).index_add_(0, idx[sel], x[sel])
        stats[k_bins * d:k_bins * d + k_bins] += torch.bincount(idx[sel], minlength=k_bins).float()

    def ema_apply(codebook, k_sum, k_elem, stats, k_rand, mu, threshold, prep=None):
        kb, d = codebook.shape
        state = orc.CodebookState(k=codebook.clone(), k_sum=k_sum.clone(), k_elem=k_elem.clone(), init=True)
        _k_sum, _k_elem = stats[:kb * d].view(kb, d), stats[kb * d:kb * d + kb]
        old_k = state.k
        state.k_sum = mu * state.k_sum + (1 - mu) * _k_sum
        state.k_elem = mu * state.k_elem + (1 - mu) * _k_elem
        usage = (state.k_elem.view(kb, 1) >= threshold).float()
        state.k = usage * (state.k_sum / state.k_elem.view(kb, 1)) + (1 - usage) * k_rand
        codebook.copy_(state.k); k_sum.copy_(state.k_sum); k_elem.copy_(state.k_elem)
        prob = _k_elem / _k_elem.sum()
        return torch.stack([-(prob * orc.safe_log(prob)).sum(), (_k_elem >= threshold).sum().float(), usage.sum(),
                            torch.norm(state.k - old_k) / np.sqrt(kb * d)]), prep
    return ema_accumulate, ema_apply


def _codebook_worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd"))
    from smt_amd import vq
    from models.vqvae.bottleneck import BottleneckBlock
    vq.ema_accumulate, vq.ema_apply = _cpu_vq_ops()
    _init(rank, world, port)
    kb, d, n = 16, 8, 50
    blk = BottleneckBlock(kb, d, 0.9, 1.0)
    g = torch.Generator().manual_seed(7)
    k0 = torch.randn(kb, d, generator=g)
    blk.init_k(None, None, k_rand=k0.clone() if rank == 0 else torch.zeros(kb, d))   # broadcast from rank 0
    assert torch.equal(blk.k, k0)
    gr = torch.Generator().manual_seed(20 + rank)
    rows = torch.randn(n, d, generator=gr)
    idx = torch.randint(0, kb // 2, (n,), generator=gr)      # half of the codes are never hit -> revival
    mask = (torch.rand(n, generator=gr) > 0.2).float()
    metrics = blk.update_k(rows, idx, mask)                    # k_rand drawn locally; rank 0's must win
    torch.save({"rows": rows, "idx": idx, "mask": mask, "k": blk.k, "k_sum": blk.k_sum, "k_elem": blk.k_elem,
                "k0": k0, "metrics": torch.stack([metrics[m] for m in ("entropy", "used_curr", "usage", "dk")])},
               os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_codebook_update_one_allreduce_two_ranks(tmp_path):
    _spawn(_codebook_worker, 2, str(tmp_path))
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(2)]
    for name in ("k", "k_sum", "k_elem", "metrics"):
        assert torch.equal(r[0][name], r[1][name]), name        # all ranks end with bit-identical state
    # oracle: reference update_k fed sum_r _k_sum_r, sum_r _k_elem_r and rank 0's revival rows
    kb, d = 16, 8
    sums, counts = torch.zeros(kb, d), torch.zeros(kb)
    for q in r:
        sel = q["mask"] != 0
        sums.index_add_(0, q["idx"][sel], q["rows"][sel])
        counts += torch.bincount(q["idx"][sel], minlength=kb).float()
    k0 = r[0]["k0"]
    k_sum = 0.9 * k0 + 0.1 * sums
    k_elem = 0.9 * torch.ones(kb) + 0.1 * counts
    assert torch.allclose(r[0]["k_sum"], k_sum, atol=1e-6) and torch.allclose(r[0]["k_elem"], k_elem, atol=1e-6)
    used = k_elem >= 1.0
    assert torch.allclose(r[0]["k"][used], (k_sum / k_elem[:, None])[used], atol=1e-6)
    dead = ~used
    assert dead.any()
    # revived codes are rows of RANK 0's data (its k_rand survived the all-reduce; others contributed zeros)
    rows0 = r[0]["rows"][r[0]["mask"] != 0]
    for row in r[0]["k"][dead]:
        assert (rows0 - row).abs().sum(1).min() < 1e-6
