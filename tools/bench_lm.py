#!/usr/bin/env python3
"""Step time of the TransformerLM train step on the reference's configuration and batch shape
(configs/models/transformer_lm.yaml; scripts/train_transformer_lm.sh: batch 8 x 258 tokens): forward + backward + fused
AdamW, synthetic codes, random-init weights.  Prints one JSON line: tokens/s and ms/step.

    python tools/bench_lm.py [--steps 20] [--warmup 5] [--batch 8] [--len 258]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "speech-masters-thesis_amd"))

import torch  # noqa: E402


def build(tmp, precision="fp32", device="cuda:0"):
    """(model, optimizer, scheduler) of the reference configuration around a throw-away VQ-VAE run directory.  fp32 is the
    only precision (bf16 projections were measured and dropped)."""
    assert precision == "fp32"
    from models.transformer_lm.transformer_lm import TransformerLM
    from utils import config as C
    from utils.commons import get_model, get_optimizer, setup_logdir
    from utils.train_utils import save_checkpoint
    pkg = os.path.join(ROOT, "speech-masters-thesis_amd")
    log_dir = os.path.join(tmp, "vqvae")
    cfg = C.merge(C.load(os.path.join(pkg, "configs/models/vqvae.yaml")), C.load(os.path.join(pkg, "configs/datasets/synthetic_ljspeech.yaml")),
                  C.create({"train": {"batch_size": 2, "n_gpus": 1, "ema": False, "log_dir": log_dir, "num_workers": 0, "total_epochs": 1}}))
    cfg.model.update(C.create(dict(width=16, emb_width=32, l_bins=512, multipliers=[1, 1, 1])))
    setup_logdir(cfg)
    vq, ema = get_model(cfg, device)
    opt, sched = get_optimizer(cfg, vq)
    save_checkpoint(cfg, 1, 0, vq, ema, opt, sched)
    lm_cfg = C.load(os.path.join(pkg, "configs/models/transformer_lm.yaml"))
    lm_cfg.model.vqvae.log_dir, lm_cfg.model.vqvae.ckpt_num = log_dir, 1
    torch.manual_seed(0)
    model = TransformerLM(lm_cfg).to(device)
    optimizer, scheduler = get_optimizer(lm_cfg, model)
    return model, optimizer, scheduler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--len", type=int, default=258)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        model, optimizer, scheduler = build(tmp)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(2, 514, (args.batch, args.len), generator=g)
    x[:, 0] = 1
    x[:, -1] = 0
    lens = torch.full((args.batch,), args.len - 1)
    x, lens = x.cuda(), lens.cuda()
    model.train()

    def step():
        optimizer.zero_grad(set_to_none=True)
        out, metrics = model(x, lens, None, None)
        out["loss"].backward()
        optimizer.step()
        scheduler.step()
        return out["loss"]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "transformer_lm_train_tokens_per_s", "value": args.batch * args.len / dt, "unit": "tokens/s",
                      "ms_per_step": dt * 1e3, "batch": args.batch, "len": args.len, "loss": float(loss.detach()),
                      "config": "transformer_lm.yaml (12 x d512 h16 ff2048, dropout 0.1)"}))


if __name__ == "__main__":
    main()
