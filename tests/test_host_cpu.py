"""Host-side mirror of the reference interfaces (no GPU): config tree, `_import_` registry, model API
slot mapping, dataset contract / collate, state-dict layout, CLI flags, checkpoint dictionary."""
import json
import os

import pytest
import torch

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd")


def load_cfg(model="vqvae", dataset="synthetic_ljspeech", **train):
    from utils import config as C
    return C.merge(C.load(os.path.join(PKG, "configs/models", model + ".yaml")),
                   C.load(os.path.join(PKG, "configs/datasets", dataset + ".yaml")),
                   C.create({"train": dict(batch_size=2, n_gpus=0, **train)}))


def test_config_tree_semantics(tmp_path):
    from utils import config as C
    cfg = load_cfg()
    assert cfg.optimizer.eps == 1e-9 and isinstance(cfg.optimizer.eps, float)       # PyYAML reads '1e-9' as str
    assert cfg.model.loss.n_ffts == [2048, 1024, 512] and cfg.get("scheduler") is None
    assert cfg.model.get("nope", 3) == 3 and cfg["model"]["l_bins"] == cfg.model.l_bins == 512
    cfg.model.levels = 1                                                           # in-place assignment
    C.save(cfg, tmp_path / "c.yaml")
    again = C.load(tmp_path / "c.yaml")
    assert again.to_dict() == cfg.to_dict()
    merged = C.merge(cfg, C.create({"model": {"l_bins": 256}}))
    assert merged.model.l_bins == 256 and merged.model.emb_width == 128 and cfg.model.l_bins == 512
    for name, k in (("vqvae_k256", 256), ("vqvae_k1024", 1024)):                   # BASELINE.json configs
        assert load_cfg(name).model.l_bins == k


def test_state_dict_layout_matches_reference_inventory():
    from models.vqvae.vqvae import VQVAE
    cfg = load_cfg()
    model = VQVAE(cfg)
    assert cfg.model.levels == 1 and cfg.model.multipliers == [1]                  # the in-place hack, vqvae.py:65-70
    with open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_inventory.json")) as f:
        inv = json.load(f)
    ref = {k: v for k, v in inv["entries"].items() if not k.endswith("_basis")}
    sd = model.state_dict()
    assert list(sd) == list(ref) and all(list(sd[k].shape) == ref[k] for k in ref)
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == inv["n_trainable"] == 7405441
    # a reference checkpoint also carries the six DFT-basis buffers: tolerated on load
    full = dict(sd)
    full["multi_stft_loss.stfts.0.forward_basis"] = torch.zeros(2050, 1, 2048)
    model.load_state_dict(full)
    # zero-initialised layers of the reference (resnet.py:29-32, :218-220)
    assert all(float(v.abs().sum()) == 0 for k, v in sd.items() if ".model.5." in k or ".gate." in k)


def test_model_api_slot_mapping():
    from models import base

    class Probe(base.WaveformReconstructionModel):
        def forward(self, x, x_lengths, speaker=None):
            return {"loss": x.sum(), "seen": (x.shape, x_lengths.tolist(), speaker)}, {}

    x = torch.randn(2, 1, 8)
    loss_dict, metrics = Probe().supervised_step(["tok", "tl", "sp", "sl", x, torch.tensor([8, 4]), "spk"])
    assert loss_dict["seen"] == (x.shape, [8, 4], "spk") and torch.equal(loss_dict["y"], x[:, 0]) and metrics == {}

    class Spec(base.TokenToSpectrogramModel):
        def forward(self, x, xl, y, yl, speaker=None):
            return {"loss": torch.zeros(()), "args": (x, xl, yl)}, {}

    out, _ = Spec().supervised_step(["tok", "tl", "sp", "sl", None, None, None])
    assert out["args"] == ("tok", "tl", "sl") and out["y"] == "sp"
    with pytest.raises(NotImplementedError):
        base.SpectrogramReconstructionModel().supervised_step([None] * 7)


def test_synthetic_dataset_contract_and_collate():
    from utils.commons import _resolve
    cfg = load_cfg()
    cfg.dataset.clip_length = 4096 + 100
    ds_cls = _resolve(cfg.dataset["_import_"])
    train, val = ds_cls(cfg, "train"), ds_cls(cfg, "val")
    assert len(val) == 10 and len(train) == cfg.dataset.num_clips
    item = train[3]
    assert len(item) == 7 and item[0] is None and item[5] == 4096 and item[4].shape == (4096,)   # cut to x512
    assert float(item[4].abs().max()) <= 1.0 and torch.equal(train[3][4], item[4])              # seeded
    cfg.dataset.ragged = True
    ragged = ds_cls(cfg, "train")
    batch = ds_cls.collate([ragged[0], ragged[1], ragged[2]])
    lens = batch[5]
    assert batch[4].shape == (3, 1, int(lens.max())) and lens.dtype == torch.long and (lens % 512 == 0).all()
    assert all(b is None for i, b in enumerate(batch) if i not in (4, 5))
    short = int(lens.argmin())
    assert float(batch[4][short, 0, int(lens[short]):].abs().sum()) == 0.0                       # zero padding


def test_ljspeech_reads_16bit_pcm_like_librosa(tmp_path):
    import wave
    import numpy as np
    from datasets.ljspeech import LJSpeech
    root = tmp_path / "LJSpeech-1.1"
    (root / "wavs").mkdir(parents=True)
    rng = np.random.default_rng(0)
    lines = []
    for i in range(12):
        pcm = rng.integers(-32768, 32767, size=1000 + 37 * i, dtype=np.int16)
        with wave.open(str(root / "wavs" / f"LJ{i:03d}.wav"), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(22050); f.writeframes(pcm.tobytes())
        lines.append(f"LJ{i:03d}|text {i}|text {i}")
    (root / "metadata.csv").write_text("\n".join(lines) + "\n")
    cfg = load_cfg(dataset="ljspeech")
    cfg.dataset.dataset_path = str(root)
    cfg.dataset.use_spect = cfg.dataset.use_token = False          # what get_model does for the VQ-VAE
    train, val = LJSpeech(cfg, "train"), LJSpeech(cfg, "val")
    assert len(val) == 10 and len(train) == 2                      # first ten clips validate (ljspeech.py:40-45)
    item = val[0]
    assert item[5] == 512 and item[4].dtype == torch.float32 and float(item[4].abs().max()) <= 1.0
    with wave.open(str(root / "wavs" / "LJ000.wav"), "rb") as f:
        ref = np.frombuffer(f.readframes(512), dtype="<i2").astype(np.float32) / 32768.0
    assert np.array_equal(item[4].numpy(), ref)
    cfg.dataset.use_token = True
    with pytest.raises(NotImplementedError):
        LJSpeech(cfg, "val")[0]


def test_cli_flags_and_config_merge(monkeypatch):
    import train
    args = train.parse_args(["--model", "vqvae_k256", "--dataset", "synthetic_ljspeech", "--batch_size", "4", "--ema",
                             "--grad_clip_norm", "1.5", "--n_gpus", "2", "--total_epochs", "3", "--load_ckpt", "x.pt",
                             "--ckpt_every_n_steps", "7", "--log_every_n_steps", "2", "--eval_every_n_epochs", "1",
                             "--run_sanity_val_epoch", "--fp16", "--seed", "5", "--num_workers", "0", "--log_dir", "/tmp/l"])
    monkeypatch.chdir(PKG)
    cfg = train.build_config(args)
    assert cfg.model.l_bins == 256 and cfg.train.batch_size == 4 and cfg.train.ema and cfg.train.grad_clip_norm == 1.5
    assert set(cfg.train) == {"log_dir", "seed", "batch_size", "ema", "grad_clip_norm", "fp16", "num_workers", "n_gpus",
                              "total_epochs", "load_ckpt", "ckpt_every_n_steps", "log_every_n_steps",
                              "eval_every_n_epochs", "run_sanity_val_epoch"}


def test_optimizer_scheduler_ema_and_checkpoint_layout(tmp_path):
    from models.ema import EMA, DummyEMA
    from utils.commons import get_optimizer
    from utils.train_utils import accumulate_stats, save_checkpoint
    from collections import defaultdict
    cfg = load_cfg(log_dir=str(tmp_path), total_epochs=2, ema=True)
    os.makedirs(tmp_path / "ckpts")
    model = torch.nn.Linear(4, 3)
    opt, sched = get_optimizer(cfg, model)
    g = opt.param_groups[0]
    assert isinstance(opt, torch.optim.AdamW) and g["lr"] == 1e-4 and g["betas"] == (0.9, 0.98) and g["eps"] == 1e-9
    ema = EMA(model, mu=0.9)
    w0 = model.weight.detach().clone()
    with torch.no_grad():
        model.weight.add_(1.0)
    ema.step()
    assert torch.allclose(ema.state_dict()["weight"], 0.9 * w0 + 0.1 * (w0 + 1), atol=1e-6)   # ema.py:55-58
    ema.swap()
    assert torch.allclose(model.weight, 0.9 * w0 + 0.1 * (w0 + 1), atol=1e-6)
    ema.swap()
    assert DummyEMA().step() is None
    model(torch.randn(2, 4)).sum().backward(); opt.step(); sched.step()
    assert opt.param_groups[0]["lr"] == 1e-4                                                   # DummyLR: constant
    path = save_checkpoint(cfg, 12, 1, model, ema, opt, sched)
    ckpt = torch.load(path, weights_only=True)
    # the reference's seven keys (train_utils.py:148-171) + "extra" (codebook accumulators / dropout counter, tolerated by loaders)
    assert set(ckpt) == {"config", "model", "optim", "sched", "ema", "step", "epoch", "extra"} and ckpt["step"] == 12
    assert os.path.basename(save_checkpoint(cfg, 99, -1, model, ema, opt, sched)) == "ckpt.last.pt"
    losses, metrics = defaultdict(float), defaultdict(float)
    accumulate_stats(2, {"loss": torch.tensor(4.0), "loss_x": torch.tensor(2.0), "yh": torch.zeros(3)},
                     {"fit": torch.tensor(1.0)}, losses, metrics)
    assert dict(losses) == {"loss": 2.0, "loss_x": 1.0} and dict(metrics) == {"fit": 0.5}


def test_training_forward_scope_marks_packed_weights_stale_once_per_pass():
    """smt_amd.convops.training_forward: the outermost scope of a TRAINING forward marks the packed operand copies stale
    (one repack per pass), nested scopes and eval-mode scopes do not; every torch optimizer's step does, whoever built it."""
    import torch
    from smt_amd import convops
    c = convops._pack_cache
    g0 = c.generation
    with convops.training_forward(True):
        with convops.training_forward(True):
            pass
        with convops.training_forward(True):
            pass
    assert c.generation == g0 + 1 and c.dirty
    with convops.training_forward(False):
        with convops.training_forward(True):      # a training sub-module inside an eval-mode parent still refreshes
            pass
    assert c.generation == g0 + 2
    with convops.training_forward(False):
        pass
    assert c.generation == g0 + 2
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    torch.optim.SGD([p], lr=0.1).step()           # an optimizer nobody registered anything on
    assert c.generation == g0 + 3

    class M(torch.nn.Module):
        @convops.forward_scope
        def forward(self, x):
            return x + 1
    m = M()
    m(torch.zeros(1))
    assert c.generation == g0 + 4
    m.eval()
    m(torch.zeros(1))
    assert c.generation == g0 + 4


def test_stale_autograd_graph_probe():
    """smt_amd.graph.stale_autograd_graphs: a parameter whose AccumulateGrad node is held by a live graph of an earlier
    iteration is reported; once that graph is gone it is not (no GPU needed: the probe runs no kernel)."""
    import torch
    from smt_amd.graph import stale_autograd_graphs
    params = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2)), torch.zeros(2)]
    assert stale_autograd_graphs(params) == []
    loss = (params[1] * 2).sum()
    assert stale_autograd_graphs(params) == [1]
    assert stale_autograd_graphs(params) == [1]           # the probe leaves no trace of its own
    del loss
    assert stale_autograd_graphs(params) == []
