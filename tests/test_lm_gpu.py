"""the TransformerLM over VQ codes through libsmt_hip.so (csrc/lm.hip) against the CPU oracle
(oracle/lm_oracle.py, pinned on the reference's own class by tests/golden/transformer_lm.npz).

Tolerances (fp32 path): kernels alone vs the float64 oracle 1e-5 absolute on O(1) values; whole-model logits 2e-4 and
gradients 2e-3 relative L2 (fp32 GEMM accumulation order through 12 layers), stated at each assert."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lm_oracle as lmo

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd")
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_embedding_kernel(p):
    from smt_amd import lm as K
    x, _ = lmo.synthetic_tokens(3, 19, 16, seed=1)
    w = torch.randn(18, 64, generator=torch.Generator().manual_seed(2))
    w[0] = 0
    pe = lmo.positional_table(32, 64)
    drop = lmo.CounterDropout(seed=7, p=p)
    wd = w.double().requires_grad_(True)
    ref = drop(0, F.embedding(x, wd) * math.sqrt(64) + pe[None, :19].double())
    dy = torch.randn(3, 19, 64, generator=torch.Generator().manual_seed(3))
    ref.backward(dy.double())
    wg = w.to(DEV).requires_grad_(True)
    out = K.embed(x.to(DEV), wg, pe.to(DEV), K.Drop(p, True, 7, 0), 0)
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu(), ref.float(), atol=1e-5)
    grad_ref = wd.grad.clone()
    grad_ref[0] = 0                                                 # padding_idx row (nn.Embedding semantics)
    assert torch.allclose(wg.grad.cpu(), grad_ref.float(), atol=1e-4)


@pytest.mark.parametrize("causal,ragged,p,b,l,h", [(True, True, 0.0, 3, 21, 2), (True, True, 0.1, 2, 70, 4), (False, False, 0.0, 2, 9, 2),
                                                  (True, False, 0.1, 1, 258, 16), (True, True, 0.0, 2, 300, 1),
                                                  (True, True, 0.1, 2, 513, 2), (False, True, 0.0, 1, 1030, 1)])
def test_attention_kernel_forward_and_backward(causal, ragged, p, b, l, h):
    from smt_amd import lm as K
    g = torch.Generator().manual_seed(l)
    qkv = torch.randn(b, l, 3 * h * 32, generator=g)
    lens = torch.randint(l // 2, l + 1, (b,), generator=g) if ragged else None
    dy = torch.randn(b, l, h * 32, generator=g)
    q64 = qkv.double().requires_grad_(True)
    ref = lmo.attention_core(q64, lens, h, causal, lmo.CounterDropout(seed=5, p=p), 3)
    ref.backward(dy.double())
    qg = qkv.to(DEV).requires_grad_(True)
    out = K.attention(qg, None if lens is None else lens.to(DEV, torch.int32), h, causal, K.Drop(p, True, 5, 3))
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu(), ref.float(), atol=2e-5), float((out.cpu() - ref.float()).abs().max())
    assert rel(qg.grad, q64.grad) < 2e-5
    assert torch.allclose(qg.grad.cpu(), q64.grad.float(), atol=5e-5)


@pytest.mark.parametrize("rows,dim,p,with_x,with_h,with_bias", [(37, 64, 0.1, True, True, True), (130, 512, 0.1, True, True, True),
                                                                 (9, 512, 0.0, True, False, False), (5, 2048, 0.0, False, True, True),
                                                                 (8, 768, 0.2, True, True, False)])
def test_add_layer_norm_kernel(rows, dim, p, with_x, with_h, with_bias):
    from smt_amd import lm as K
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, dim, generator=g) if with_x else None
    h = torch.randn(rows, dim, generator=g) * 2 if with_h else None
    hb = torch.randn(dim, generator=g) if with_bias else None
    gamma, beta = 1 + 0.2 * torch.randn(dim, generator=g), 0.2 * torch.randn(dim, generator=g)
    gamma[3] = 0.0                                                  # a dead scale: its xhat must still reach dgamma
    dy = torch.randn(rows, dim, generator=g)
    drop = lmo.CounterDropout(seed=9, p=p)
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, h, gamma, beta, hb)]
    pre = (leaves[0] if with_x else 0) + (drop(2, leaves[1] + (leaves[4] if with_bias else 0)) if with_h else 0)
    ref = F.layer_norm(pre, (dim,), leaves[2], leaves[3], 1e-5)
    ref.backward(dy.double())
    dev = [t.to(DEV).requires_grad_(True) if t is not None else None for t in (x, h, gamma, beta, hb)]
    out = K.add_layer_norm(dev[0], dev[1], dev[2], dev[3], 1e-5, K.Drop(p, True, 9, 2), h_bias=dev[4])
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu(), ref.float(), atol=2e-5)
    for mine, want in zip(dev, leaves):
        if mine is not None:
            assert torch.allclose(mine.grad.cpu(), want.grad.float(), atol=1e-4, rtol=1e-4), float((mine.grad.cpu() - want.grad.float()).abs().max())


def test_add_layer_norm_rejects_an_unbuilt_width():
    from smt_amd import lm as K
    x = torch.randn(4, 320, device=DEV)
    with pytest.raises(RuntimeError, match="not built"):
        K.add_layer_norm(x, None, torch.ones(320, device=DEV), torch.zeros(320, device=DEV))


@pytest.mark.parametrize("rows,dim,p", [(50, 128, 0.1), (2064, 2048, 0.1), (7, 256, 0.0)])
def test_bias_relu_dropout_kernel(rows, dim, p):
    from smt_amd import lm as K
    g = torch.Generator().manual_seed(rows)
    h, bias, dy = torch.randn(rows, dim, generator=g), torch.randn(dim, generator=g), torch.randn(rows, dim, generator=g)
    h64, b64 = h.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = lmo.CounterDropout(seed=4, p=p)(6, F.relu(h64 + b64))
    ref.backward(dy.double())
    hg, bg = h.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    out = K.bias_relu_dropout_(hg * 1.0, bg, K.Drop(p, True, 4, 6))           # * 1.0: the op works in place on a non-leaf
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu(), ref.float(), atol=1e-6)
    assert torch.allclose(hg.grad.cpu(), h64.grad.float(), atol=1e-6)
    assert torch.allclose(bg.grad.cpu(), b64.grad.float(), atol=1e-3, rtol=1e-5)


def test_cross_entropy_kernel_masks_rows_and_breaks_ties_low():
    from smt_amd import lm as K
    g = torch.Generator().manual_seed(8)
    logits = torch.randn(41, 512, generator=g) * 3
    target = torch.randint(0, 512, (41,), generator=g)
    target[::5] = -1
    logits[1, 7] = logits[1, 300] = 50.0                               # a tie: torch.argmax takes the lower index
    target[1] = 7
    keep = target >= 0
    l64 = logits.double().requires_grad_(True)
    ref = F.cross_entropy(l64[keep], target[keep])
    ref.backward()
    acc_ref = (l64[keep].argmax(1) == target[keep]).double().mean()
    lg = logits.to(DEV).requires_grad_(True)
    loss, acc, count = K.cross_entropy(lg, target.to(DEV))
    (loss * 1.5).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 and abs(acc.item() - acc_ref.item()) < 1e-7 and int(count) == int(keep.sum())
    assert torch.allclose(lg.grad.cpu(), 1.5 * l64.grad.float(), atol=1e-7, rtol=1e-4)
    # no scored row at all: nan like the reference's mean over an empty selection
    loss, acc, _ = K.cross_entropy(lg.detach(), torch.full((41,), -1, device=DEV))
    assert math.isnan(float(loss)) and math.isnan(float(acc))


# ------------------------------------------------------------------------------------------------ model
def _vqvae_run(tmp_path, l_bins=16):
    """A small VQ-VAE run directory (config.yaml + ckpts/ckpt.3.pt) for TransformerLM.load_vqvae."""
    from utils import config as C
    from utils.commons import get_model, get_optimizer, setup_logdir
    from utils.train_utils import save_checkpoint
    log_dir = str(tmp_path / "vqvae")
    cfg = C.merge(C.load(os.path.join(PKG, "configs/models/vqvae.yaml")),
                  C.load(os.path.join(PKG, "configs/datasets/synthetic_ljspeech.yaml")),
                  C.create({"train": {"batch_size": 2, "n_gpus": 1, "ema": False, "log_dir": log_dir, "num_workers": 0, "total_epochs": 1}}))
    cfg.model.update(C.create(dict(width=16, emb_width=32, l_bins=l_bins, multipliers=[1, 1, 1])))
    torch.manual_seed(0)
    setup_logdir(cfg)
    model, ema = get_model(cfg, DEV)
    opt, sched = get_optimizer(cfg, model)
    model.bottleneck.level_blocks[0].k.copy_(torch.randn(l_bins, 32, generator=torch.Generator().manual_seed(1)).to(DEV) * 0.3)
    save_checkpoint(cfg, 3, 0, model, ema, opt, sched)
    return log_dir, model


def _lm_config(log_dir, **over):
    from utils import config as C
    cfg = C.load(os.path.join(PKG, "configs/models/transformer_lm.yaml"))
    cfg.model.update(C.create(over))
    cfg.model.vqvae.log_dir, cfg.model.vqvae.ckpt_num = log_dir, 3
    return cfg


def _build(tmp_path, params=None, **over):
    from models.transformer_lm.transformer_lm import TransformerLM
    log_dir, vq = _vqvae_run(tmp_path, l_bins=over.get("vocab_size", 512))
    torch.manual_seed(0)
    model = TransformerLM(_lm_config(log_dir, **over)).to(DEV)
    if params is not None:
        missing, unexpected = model.load_state_dict({k: v for k, v in params.items()}, strict=False)
        assert not unexpected and all(k.startswith("vqvae.") or k == "pos_encoding.pe" for k in missing), (missing, unexpected)
    return model, vq


SMALL = dict(vocab_size=16, embed_dim=64, max_len=64, num_layers=2, d_model=64, nhead=2, dim_feedforward=128, dropout=0.0)


def test_small_model_matches_the_reference_golden(golden, tmp_path):
    """The reference's own TransformerLM (tests/golden/transformer_lm.npz): same parameters, same batch -> its loss,
    accuracy and gradients through the HIP path.  fp32 both sides: logits 1e-4 abs, loss 1e-5, gradients 1e-3 rel-L2."""
    g = golden("transformer_lm")
    params = {k[len("param."):]: torch.from_numpy(g[k]) for k in g if k.startswith("param.")}
    model, _ = _build(tmp_path, params, **SMALL)
    # state_dict surface: exactly the reference's names for everything the fixture holds
    assert set(params) <= set(model.state_dict()) and "pos_encoding.pe" in model.state_dict()
    assert model.state_dict()["pos_encoding.pe"].shape == (64, 1, 64)
    x, lens = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["lens"]).to(DEV)
    model.train()
    logits = model.logits(x, lens.to(torch.int32))
    assert torch.allclose(logits.cpu(), torch.from_numpy(g["logits"]), atol=1e-4)
    loss_dict, metrics = model(x, lens, None, None)
    assert loss_dict["yh"] is None
    assert abs(float(loss_dict["loss"]) - float(g["loss"])) < 1e-5 and abs(float(metrics["accuracy"]) - float(g["accuracy"])) < 1e-7
    loss_dict["loss"].backward()
    for name, prm in model.named_parameters():
        if name.startswith("vqvae."):
            assert prm.grad is None
            continue
        ref = torch.from_numpy(g["grad." + name])
        assert rel(prm.grad, ref) < 1e-3, (name, rel(prm.grad, ref))


def test_small_model_train_mode_with_dropout_matches_the_oracle(tmp_path):
    """Dropout on (p = 0.1) at every site: the product's masks are restated by the oracle, so the whole train-mode forward and
    backward is comparable.  float64 oracle; fp32 product: loss 2e-5, gradients 1e-3 rel-L2."""
    model, _ = _build(tmp_path, **dict(SMALL, dropout=0.1, num_layers=3))
    p32 = lmo.init_params(16, 64, 2, 128, 3, seed=21)
    model.load_state_dict(p32, strict=False)
    x, lens = lmo.synthetic_tokens(4, 33, 16, seed=22)
    model.train()
    model._drop_seed = 40
    loss_dict, metrics = model(x.to(DEV), lens.to(DEV), None, None)
    loss_dict["loss"].backward()
    p64 = {k: v.double().requires_grad_(True) for k, v in p32.items()}
    logits = lmo.lm_logits(x, lens, p64, heads=2, num_layers=3, drop=lmo.CounterDropout(seed=41, p=0.1))   # forward bumps the seed
    loss, acc = lmo.lm_loss(x, logits)
    loss.backward()
    assert abs(float(loss_dict["loss"]) - float(loss)) < 2e-5 and abs(float(metrics["accuracy"]) - float(acc)) < 1e-6
    for name, prm in model.named_parameters():
        if not name.startswith("vqvae."):
            assert rel(prm.grad, p64[name].grad) < 1e-3, (name, rel(prm.grad, p64[name].grad))
    # a second step draws different masks (the counter advanced)
    again, _ = model(x.to(DEV), lens.to(DEV), None, None)
    assert abs(float(again["loss"]) - float(loss_dict["loss"])) > 1e-6


def test_eval_reconstructs_audio_from_the_argmax_codes_and_sample_runs(tmp_path):
    model, vq = _build(tmp_path, **SMALL)
    x, lens = lmo.synthetic_tokens(2, 12, 16, seed=5)
    model.eval()
    vq.eval()
    out, metrics = model(x.to(DEV), lens.to(DEV), None, None)
    codes = model.logits(x.to(DEV), lens.to(DEV, torch.int32))[:, :-1].argmax(-1)
    want = vq.dequantize_and_decode(codes, torch.minimum(lens, torch.tensor(11)).to(DEV))[:, 0]
    assert out["yh"].shape == (2, 11 * 128) and torch.equal(out["yh"], want.float())
    if int(lens[1]) < 11:
        assert float(out["yh"][1, int(lens[1]) * 128:].abs().max()) == 0.0        # beyond the sequence: silence
    torch.manual_seed(0)
    audio, q = model.sample(batch_size=3, n_steps=6, device=DEV, sigma=1.0)
    assert q.shape == (3, 6) and int(q.min()) >= 0 and int(q.max()) < 16 and audio.shape == (3, 6 * 128)
    assert torch.isfinite(audio).all()
    # the script's default --n_steps (1024, as in the reference) walks prefixes longer than 512 tokens
    long_model, _ = _build(tmp_path / "long", **{**SMALL, "max_len": 600})
    long_model.eval()
    audio, q = long_model.sample(batch_size=1, n_steps=520, device=DEV, sigma=1.0)
    assert q.shape == (1, 520) and audio.shape == (1, 520 * 128) and torch.isfinite(audio).all()


def test_sample_step_is_the_uncausal_forward_of_the_reference(tmp_path):
    """transformer_lm.py:142: `sample` runs the encoder with mask=None -- every prefix position sees the whole prefix."""
    model, _ = _build(tmp_path, **SMALL)
    p32 = lmo.init_params(16, 64, 2, 128, 2, seed=31)
    model.load_state_dict(p32, strict=False)
    model.eval()
    x, _ = lmo.synthetic_tokens(2, 9, 16, seed=6, ragged=False)
    with torch.no_grad():
        got = model.logits(x.to(DEV), None, causal=False)
    want = lmo.lm_logits(x, None, {k: v.double() for k, v in p32.items()}, heads=2, num_layers=2, causal=False)
    assert torch.allclose(got.cpu(), want.float(), atol=1e-4)


def test_alternative_losses_match_their_reference_formulas(tmp_path):
    """loss_type mmi / focal (reference models/transformer_lm/losses.py) on the scored rows of the same logits."""
    from models.transformer_lm.losses import FocalLoss, MaximumMutualInformationLoss
    g = torch.Generator().manual_seed(3)
    yh, y = torch.randn(29, 16, generator=g).double(), torch.randint(0, 16, (29,), generator=g)
    p = F.softmax(yh, -1)
    pz = p.mean(0)
    ref_mmi = -(p * F.log_softmax(F.one_hot(y, 16).double(), -1)).sum(-1).mean(0) + (pz * pz.log()).sum(-1)
    assert abs(float(MaximumMutualInformationLoss(16)(yh, y)) - float(ref_mmi)) < 1e-12
    logp = F.log_softmax(yh, -1)[torch.arange(29), y]
    ref_focal = ((1 - logp.exp()) ** 10.0 * -logp).mean()
    assert abs(float(FocalLoss(gamma=10.0)(yh, y)) - float(ref_focal)) < 1e-12
    model, _ = _build(tmp_path, **dict(SMALL, loss_type="focal"))
    x, lens = lmo.synthetic_tokens(2, 12, 16, seed=5)
    model.train()
    out, metrics = model(x.to(DEV), lens.to(DEV), None, None)
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and model.classifier.weight.grad.abs().sum() > 0 and 0 <= float(metrics["accuracy"]) <= 1


def test_full_size_train_step_matches_the_oracle(tmp_path):
    """The configuration the reference trains (configs/models/transformer_lm.yaml: 12 layers, d 512, 16 heads, ff 2048,
    dropout 0.1) on its batch shape (scripts/train_transformer_lm.sh: batch 8, 256 codes + <bos> + pad = 258), against the
    fp32 CPU oracle with the same dropout masks.  Post-norm keeps activations O(1): logits 5e-4 rel-L2, loss 1e-4,
    gradients 5e-3 rel-L2 (fp32 GEMMs, different accumulation orders)."""
    model, _ = _build(tmp_path)
    p32 = lmo.init_params(512, 512, 16, 2048, 12, seed=51)
    model.load_state_dict(p32, strict=False)
    x, lens = lmo.synthetic_tokens(8, 258, 512, seed=52)
    model.train()
    model._drop_seed = 6
    loss_dict, metrics = model(x.to(DEV), lens.to(DEV), None, None)
    loss_dict["loss"].backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    pc = {k: v.clone().requires_grad_(True) for k, v in p32.items()}
    logits = lmo.lm_logits(x, lens, pc, heads=16, num_layers=12, drop=lmo.CounterDropout(seed=7, p=0.1))
    loss, acc = lmo.lm_loss(x, logits)
    loss.backward()
    assert abs(float(loss_dict["loss"]) - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    assert abs(float(metrics["accuracy"]) - float(acc)) < 2e-3                       # a near-tie may flip one of ~2000 rows
    worst = max((rel(prm.grad, pc[name].grad), name) for name, prm in model.named_parameters() if not name.startswith("vqvae."))
    assert worst[0] < 5e-3, worst


def test_train_py_end_to_end_on_generated_codes(tmp_path, monkeypatch):
    """The reference's pipeline for this model (scripts/train_transformer_lm.sh): VQ-VAE checkpoint ->
    scripts/generate_vq_dataset.py -> `train.py --model transformer_lm --dataset vqlatent`; two epochs, validation with
    reconstructed audio, checkpoint written and resumed."""
    import train
    from scripts import generate_vq_dataset as G
    from utils import config as C
    log_vq, _ = _vqvae_run(tmp_path, l_bins=16)
    dump_dir = str(tmp_path / "VQ-Latent")
    ds_cfg = C.load(os.path.join(log_vq, "config.yaml"))
    ds_cfg.dataset.update(C.create(dict(num_clips=6, ragged=True)))
    C.save(ds_cfg, os.path.join(log_vq, "config.yaml"))
    G.main(["--log_dir", log_vq, "--ckpt_num", "3", "--dump_dir", dump_dir, "--batch_size", "3", "--n_processes", "1", "--n_workers", "0"])
    monkeypatch.chdir(PKG)
    C.save(_lm_config(log_vq, **SMALL, ), "configs/models/_test_lm.yaml")
    vql = C.load("configs/datasets/vqlatent.yaml")
    vql.dataset.update(C.create(dict(dataset_path=dump_dir, segment_length=24)))
    C.save(vql, "configs/datasets/_test_vql.yaml")
    try:
        log_dir = str(tmp_path / "run")
        argv = ["--model", "_test_lm", "--dataset", "_test_vql", "--batch_size", "3", "--num_workers", "0", "--total_epochs", "2",
                "--log_every_n_steps", "1", "--ckpt_every_n_steps", "2", "--eval_every_n_epochs", "1", "--log_dir", log_dir,
                "--n_gpus", "1", "--run_sanity_val_epoch"]
        train.main(argv)
        last = torch.load(os.path.join(log_dir, "ckpts", "ckpt.last.pt"), weights_only=True)
        assert last["step"] == 4 and "transformer.layers.1.self_attn.in_proj_weight" in last["model"]
        assert "vqvae.bottleneck.k" in last["model"] and last["extra"]["drop_seed"] == 4
        import json
        import wave
        scal = os.path.join(log_dir, "scalars.jsonl")
        if os.path.exists(scal):
            tags = {json.loads(line)["tag"] for line in open(scal)}
            assert {"loss/train_loss", "metrics/train_accuracy", "loss/val_loss"} <= tags
        with wave.open(os.path.join(log_dir, "audio", "val_audio_2_pred.wav")) as w:
            assert w.getframerate() == 22050 and w.getnframes() >= 24 * 128
        train.main(argv + ["--load_ckpt", os.path.join(log_dir, "ckpts", "ckpt.last.pt")])
        # scripts/sample_from_lm.py on the run just written: wavs, the spectrogram image, the token table
        from scripts import sample_from_lm as S
        out = S.main(["--log_dir", log_dir, "--ckpt_num", "2", "--dump_dir", str(tmp_path / "outputs"), "--n_samples", "2",
                      "--n_steps", "8"])
        assert out.endswith("TransformerLM@2")
        with wave.open(os.path.join(out, "sample_1.wav")) as w:
            assert w.getnframes() == 8 * 128
        assert open(os.path.join(out, "mel_spectrograms.png"), "rb").read(8) == b"\x89PNG\r\n\x1a\n"
        table = open(os.path.join(out, "tokens.txt")).read().splitlines()
        assert len(table) == 4 and len(table[2].split()) == 8 and all(0 <= int(v) < 16 for v in table[2].split())
    finally:
        os.remove("configs/models/_test_lm.yaml")
        os.remove("configs/datasets/_test_vql.yaml")


# ------------------------------------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("b,l,h,lens", [(1, 1, 1, None), (2, 1, 2, [1, 0]), (3, 33, 2, [33, 0, 1]), (2, 64, 1, [64, 32]),
                                        (1, 512, 1, [300]), (2, 97, 3, [5, 97])])
def test_attention_edge_shapes(b, l, h, lens):
    """One token, empty sequences (lens = 0: no visible key -> zero context, zero gradient), block-aligned and maximum
    lengths, a head count that is not a power of two."""
    from smt_amd import lm as K
    g = torch.Generator().manual_seed(100 + l)
    qkv = torch.randn(b, l, 3 * h * 32, generator=g)
    dy = torch.randn(b, l, h * 32, generator=g)
    lens_t = None if lens is None else torch.tensor(lens)
    q64 = qkv.double().requires_grad_(True)
    if lens_t is not None and (lens_t == 0).any():
        # the oracle's softmax over an all-masked row is nan (as torch's own): compare the non-empty items, demand zeros
        keep = lens_t > 0
        ref = lmo.attention_core(q64[keep], lens_t[keep], h, True, lmo.CounterDropout(), 0)
        ref.backward(dy[keep].double())
    else:
        keep = torch.ones(b, dtype=torch.bool)
        ref = lmo.attention_core(q64, lens_t, h, True, lmo.CounterDropout(), 0)
        ref.backward(dy.double())
    qg = qkv.to(DEV).requires_grad_(True)
    out = K.attention(qg, None if lens_t is None else lens_t.to(DEV, torch.int32), h, True)
    out.backward(dy.to(DEV))
    assert torch.allclose(out.cpu()[keep], ref.float(), atol=3e-5)
    assert torch.allclose(qg.grad.cpu()[keep], q64.grad.float()[keep], atol=1e-4)
    assert float(out.detach().cpu()[~keep].abs().sum()) == 0.0 and float(qg.grad.cpu()[~keep].abs().sum()) == 0.0


def test_kernels_accept_empty_batches_and_reject_bad_shapes():
    from smt_amd import lm as K
    from smt_amd import native as N
    e = torch.empty(0, 7, 3 * 64, device=DEV)
    assert K.attention(e, None, 2).shape == (0, 7, 64)
    assert K.add_layer_norm(torch.empty(0, 64, device=DEV), None, torch.ones(64, device=DEV), torch.zeros(64, device=DEV)).shape == (0, 64)
    with pytest.raises(AssertionError, match="head dim 32"):
        K.attention(torch.randn(1, 4, 3 * 48, device=DEV), None, 1)
    assert K.attention(torch.randn(1, 513, 3 * 32, device=DEV), None, 1).shape == (1, 513, 32)      # no 512-token limit
    assert N.lib().smt_lm_add_ln_bwd_workspace_bytes(0, 512) == 0


def test_forward_with_no_scored_position_is_nan_like_the_reference(tmp_path):
    """All targets are pads / specials: the reference's mean over an empty selection is nan (transformer_lm.py:126-128)."""
    model, _ = _build(tmp_path, **SMALL)
    x = torch.zeros(2, 9, dtype=torch.long)
    x[:, 0] = lmo.BOS
    model.train()
    out, metrics = model(x.to(DEV), torch.tensor([1, 1]).to(DEV), None, None)
    assert math.isnan(float(out["loss"])) and math.isnan(float(metrics["accuracy"]))


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_causality_padding_and_batch_independence(tmp_path):
    """Size-independent properties at the reference's configuration (12 x d512, batch 8 x 258), eval mode, bit-exact:
    (1) causality -- tokens after position t cannot change the logits up to t; (2) key padding -- tokens at or beyond
    lens[b] cannot change the logits before lens[b], nor the loss; (3) a batch item's logits do not depend on its batch mates
    or its slot (to 2e-5: library GEMMs).  Masked scores enter the softmax as exactly zero weights, so (1) and (2) hold bit
    for bit, not to a tolerance."""
    model, _ = _build(tmp_path)
    p32 = lmo.init_params(512, 512, 16, 2048, 12, seed=61)
    model.load_state_dict(p32, strict=False)
    model.eval()
    x, lens = lmo.synthetic_tokens(8, 258, 512, seed=62)
    xd, ld = x.to(DEV), lens.to(DEV, torch.int32)
    with torch.no_grad():
        base = model.logits(xd, ld)
        # (1) change every token after position 100 (keeping them real codes)
        x2 = xd.clone()
        x2[:, 101:] = (x2[:, 101:] + 7 - lmo.OFFSET) % 512 + lmo.OFFSET
        assert torch.equal(model.logits(x2, ld)[:, :101], base[:, :101])
        # (2) scribble over the padding
        x3 = xd.clone()
        pad = torch.arange(258, device=DEV)[None, :] >= ld[:, None]
        x3[pad] = 5
        out3 = model.logits(x3, ld)
        for b in range(8):
            n = int(lens[b])
            assert torch.equal(out3[b, :n], base[b, :n])
        # (3) reverse the batch, and run one item alone: equal up to the library GEMM's row tiling (hipBLASLt picks its
        # accumulation order by row position and problem size), our kernels treat every (batch, head) alike
        rev = torch.arange(7, -1, -1, device=DEV)
        assert torch.allclose(model.logits(xd[rev], ld[rev])[rev], base, atol=2e-5)
        assert torch.allclose(model.logits(xd[3:4], ld[3:4])[0], base[3], atol=2e-5)
    loss_a, _ = model(xd, lens.to(DEV), None, None)
    x4 = xd.clone()
    x4[pad] = lmo.PAD                                    # what the dataset writes there
    loss_b, _ = model(x4, lens.to(DEV), None, None)
    assert torch.equal(loss_a["loss"], loss_b["loss"])


def test_full_size_attention_is_linear_in_the_values():
    """softmax(QK^T) V is linear in V: attention(q, k, a v1 + v2) == a attention(q, k, v1) + attention(q, k, v2) up to f32
    rounding, at 8 x 16 heads x 258 with ragged lengths and dropout ON (the mask depends on positions only)."""
    from smt_amd import lm as K
    g = torch.Generator().manual_seed(5)
    b, l, h = 8, 258, 16
    qk = torch.randn(b, l, 2 * h * 32, generator=g).to(DEV)
    v1, v2 = torch.randn(b, l, h * 32, generator=g).to(DEV), torch.randn(b, l, h * 32, generator=g).to(DEV)
    lens = torch.randint(100, l + 1, (b,), generator=g).to(DEV, torch.int32)
    drop = K.Drop(0.1, True, 9, 4)

    def att(v):
        return K.attention(torch.cat([qk, v], dim=-1), lens, h, True, drop)

    lhs, rhs = att(2.0 * v1 + v2), 2.0 * att(v1) + att(v2)
    assert torch.allclose(lhs, rhs, atol=2e-5) and float(lhs.abs().max()) > 0.1
    # scaling V by a power of two scales the context exactly
    assert torch.equal(att(4.0 * v1), 4.0 * att(v1))


# ------------------------------------------------------------------------------------------------ device keys, graph
def test_device_resident_keys_and_graphed_step_match_the_eager_step(tmp_path):
    """Keys by value, keys from device memory and a hipGraph replay must draw the SAME masks step after step: three
    training steps each way from the same weights and counter -> the same losses (bit-identical on the first step, to an
    ulp-sized drift afterwards); fresh masks on every replay."""
    from smt_amd.graph import GraphedTrainStep
    from utils.commons import get_optimizer
    x, lens = lmo.synthetic_tokens(4, 60, 16, seed=8)
    xd, ld = x.to(DEV), lens.to(DEV)
    p32 = lmo.init_params(16, 64, 2, 128, 2, seed=81)

    def fresh():
        model, _ = _build(tmp_path, **dict(SMALL, dropout=0.1))
        model.load_state_dict(p32, strict=False)
        model.train()
        model._drop_seed = 100
        cfg = _lm_config(str(tmp_path / "vqvae"), **dict(SMALL, dropout=0.1))
        opt, sched = get_optimizer(cfg, model)
        return model, opt, sched

    def eager(device_keys):
        model, opt, sched = fresh()
        if device_keys:
            model.enable_device_keys(True)
        out = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = model(xd, ld, None, None)[0]["loss"]
            loss.backward()
            opt.step(); sched.step()
            out.append(float(loss.detach()))
        return out, model._drop_seed

    def same(a, b):
        # step 1 starts from identical weights: identical masks -> bit-identical loss.  Later steps inherit the embedding
        # gradient's f32 atomics (token collisions add in arrival order), which move the weights by an ulp from run to run
        return a[0] == b[0] and all(abs(u - v) <= 2e-6 * abs(v) for u, v in zip(a, b))

    by_value, seed_a = eager(False)
    from_device, seed_b = eager(True)
    assert same(by_value, from_device) and seed_a == seed_b == 103
    assert len(set(by_value)) == 3                                   # masks (and weights) change from step to step

    model, opt, sched = fresh()
    graphed = GraphedTrainStep(model, opt, sched, xd, ld, warmup=2)  # two warm-up forward/backward passes: counter 100 -> 102
    assert model._drop_seed == 102 and int(model._seed_dev) == 102
    # the graph continues from counter 102 with untouched weights: compare with an eager run brought to the same state
    ref_model, ref_opt, ref_sched = fresh()
    ref_model._drop_seed = 102
    ref = []
    for _ in range(3):
        ref_opt.zero_grad(set_to_none=True)
        loss = ref_model(xd, ld, None, None)[0]["loss"]
        loss.backward()
        ref_opt.step(); ref_sched.step()
        ref.append(float(loss.detach()))
    got = [float(graphed.step(xd, ld)) for _ in range(3)]
    assert same(got, ref), (got, ref)
    assert model._drop_seed == 105 and int(model._seed_dev) == 105
