"""train.py end to end on the GPU with the synthetic dataset: registry, DataLoader + collate, train loop,
logging, checkpoint write / resume (reference train.py:146-221, 305-383, utils/train_utils.py:148-171)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd")


def test_train_two_epochs_checkpoint_and_resume(tmp_path, monkeypatch):
    import train
    from utils import config as C
    monkeypatch.chdir(PKG)
    small = C.load("configs/models/vqvae.yaml")
    small.model.update(C.create(dict(width=16, emb_width=32, l_bins=64, multipliers=[1, 1, 1])))
    small.model.loss.linf_topk = 128
    C.save(small, "configs/models/_test_small.yaml")
    ds = C.load("configs/datasets/synthetic_ljspeech.yaml")
    ds.dataset.update(C.create(dict(num_clips=6, clip_length=4096, ragged=False)))
    C.save(ds, "configs/datasets/_test_small.yaml")
    try:
        log_dir = str(tmp_path / "run")
        argv = ["--model", "_test_small", "--dataset", "_test_small", "--batch_size", "2", "--num_workers", "0",
                "--total_epochs", "2", "--log_every_n_steps", "1", "--ckpt_every_n_steps", "2",
                "--eval_every_n_epochs", "1", "--log_dir", log_dir, "--ema", "--n_gpus", "1", "--run_sanity_val_epoch"]
        train.main(argv)
        assert os.path.exists(os.path.join(log_dir, "config.yaml"))
        last = torch.load(os.path.join(log_dir, "ckpts", "ckpt.last.pt"), weights_only=True)
        assert last["step"] == 6 and last["epoch"] == 2                   # 6 clips / batch 2 = 3 steps x 2 epochs
        assert "bottleneck.level_blocks.0.k" in last["model"] and last["config"]["model"]["levels"] == 1
        assert os.path.exists(os.path.join(log_dir, "ckpts", "ckpt.2.pt"))
        scal = os.path.join(log_dir, "scalars.jsonl")
        if os.path.exists(scal):                                           # JSON-lines writer when tensorboard is absent
            tags = {json.loads(line)["tag"] for line in open(scal)}
            assert {"loss/train_loss", "metrics/train_fit", "loss/val_loss"} <= tags
        # resume: total_epochs reached -> no further steps, but the load path runs
        train.main(argv + ["--load_ckpt", os.path.join(log_dir, "ckpts", "ckpt.last.pt")])
    finally:
        os.remove("configs/models/_test_small.yaml")
        os.remove("configs/datasets/_test_small.yaml")
