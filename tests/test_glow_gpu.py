"""GlowTTS on the GPU (BASELINE.json configs[4]) against the reference's own fixture and against the oracle:
text encoder (relative-position attention), flow decoder (ActNorm, InvConvNear, coupling blocks with the WN stack), the
alignment search on the device, MLE / duration losses, every parameter gradient; dropout ON against the oracle with the
product's counter-based masks replayed; train.py end to end on synthetic token / mel pairs."""
import os

import numpy as np
import pytest
import torch

from oracle import glow_oracle as go

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd")
DEV = "cuda"


def _config(enc, dec, n_mels):
    from utils import config as C
    return C.create({"model": dict(n_speakers=1, gin_channels=0, encoder=dict(enc), decoder=dict(dec)),
                     "dataset": dict(n_mels=n_mels, intersperse_blanks=False, cmudict_path="")})


def _build(cfg_dict, n_mels, params):
    from models.glow_tts.glow_tts import GlowTTS
    model = GlowTTS(_config(cfg_dict["encoder"], cfg_dict["decoder"], n_mels)).to(DEV)
    missing, unexpected = model.load_state_dict({k: v.float() for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    return model


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_glow_tts_matches_the_reference_golden(golden):
    """The reference's own GlowTTS (tests/golden/glow_tts.npz: same parameters, same batch, dropout 0): losses, alignment,
    latent, log-determinants, every parameter gradient, and the eval-mode reconstruction with the captured noise.
    fp32 on both sides: losses 1e-5, z_dec 1e-4, gradients 2e-3 relative L2 per tensor (MFMA fp32 summation order)."""
    g = golden("glow_tts")
    params = {k[len("param."):]: torch.from_numpy(g[k]) for k in g if k.startswith("param.")}
    model = _build(go.GOLDEN_CFG, 8, params)
    model.encoder.pre.p_dropout = 0.0        # the prenet's dropout is 0.1 whatever the config says (modules.py:62); the fixture switched it off
    assert set(model.state_dict()) == set(params)                       # the reference's state-dict names, all of them
    tokens, x_lens = torch.from_numpy(g["tokens"]).to(DEV), torch.from_numpy(g["x_lens"]).to(DEV)
    y, y_lens = torch.from_numpy(g["y"]).to(DEV), torch.from_numpy(g["y_lens"]).to(DEV)
    model.train()
    loss_dict, metrics = model.supervised_step([tokens, x_lens, y, y_lens, None, None, None])
    assert metrics == {} and loss_dict["yh"] is None and torch.equal(loss_dict["y"], y)
    loss_dict["loss"].backward()
    assert np.isclose(loss_dict["loss_mle"].item(), float(g["loss_mle"]), rtol=1e-5), (loss_dict["loss_mle"].item(), float(g["loss_mle"]))
    assert np.isclose(loss_dict["loss_length"].item(), float(g["loss_length"]), rtol=1e-5)
    worst = (0.0, "")
    gmax = max(float(np.linalg.norm(g[k])) for k in g if k.startswith("grad."))
    for name, prm in model.named_parameters():
        assert prm.grad is not None, name
        ref = torch.from_numpy(g["grad." + name]).double()
        e = float((prm.grad.double().cpu() - ref).norm() / (ref.norm() + 1e-6 * gmax))     # floor: see the key-bias note below
        worst = max(worst, (e, name))
        assert e <= 2e-3, (name, e)
    print(f"\n[glow_tts golden] loss_mle {loss_dict['loss_mle'].item():.6f} (ref {float(g['loss_mle']):.6f}); worst gradient rel-L2 {worst}")
    # eval: reconstruction with the reference's noise
    model.eval()
    with torch.no_grad():
        ev, _ = model(tokens, x_lens, y, y_lens, noise=torch.from_numpy(g["eval_noise"]).to(DEV))
    assert ev["yh"].shape == g["eval_yh"].shape and torch.allclose(ev["yh"].cpu(), torch.from_numpy(g["eval_yh"]), atol=2e-4)
    assert np.isclose(ev["loss_mle"].item(), float(g["loss_mle"]), rtol=1e-5)


def test_alignment_pieces_match_the_fixture(golden):
    """The device-side alignment chain on the fixture's tensors: prior log-likelihood -> smt_maximum_path -> frame index ->
    gather equals the reference's path and its `torch.matmul(x_m, attn)`."""
    from models.glow_tts import submodules as S
    from smt_amd import glow
    g = golden("glow_tts")
    params = {k[len("param."):]: torch.from_numpy(g[k]).double() for k in g if k.startswith("param.")}
    tokens, x_lens = torch.from_numpy(g["tokens"]), torch.from_numpy(g["x_lens"])
    x_m, x_logs, _, _ = go.text_encoder(tokens, x_lens, params, go.GOLDEN_CFG, go.no_dropout)
    z = torch.from_numpy(g["z_dec"]).double()
    xm_c, xl_c, z_c = (t.transpose(1, 2).contiguous().float().to(DEV) for t in (x_m, x_logs, z))
    logp = glow.prior_logp(xm_c, xl_c, z_c)
    y_lens = (torch.from_numpy(g["y_lens"]) // 2) * 2
    mask = (go.sequence_mask(x_lens, logp.shape[1]).unsqueeze(-1) & go.sequence_mask(y_lens, logp.shape[2]).unsqueeze(1)).float().to(DEV)
    path = S.maximum_path(logp, mask)
    assert np.array_equal(path.cpu().numpy(), g["attn"])
    idx, dur = glow.align_index(path)
    assert torch.allclose(dur.cpu(), torch.from_numpy(g["attn"]).sum(-1))
    zm = glow.align_gather(xm_c, idx)
    assert torch.allclose(zm.cpu().transpose(1, 2), (x_m @ torch.from_numpy(g["attn"]).double()).float(), atol=1e-6)


def _masks_for(model, seed, p_by_site):
    """DropFn replaying the product's counter-based masks for the oracle's NCT tensors (the product's index is the linear
    index of the channels-last tensor; attention probabilities are [B, heads, T, T] on both sides)."""
    from oracle import vqvae_oracle as orc
    ids = model.dropout_sites()

    def drop(site, x):
        p = p_by_site(site)
        if p <= 0:
            return x
        if site.endswith(".drop") and ".attn_layers." in site:            # probabilities: same layout on both sides
            n = x.numel()
            keep = orc.dropout_keep_ntc(seed, ids[site], 1, n, 1, p).reshape(tuple(x.shape))
        else:
            b, c, t = x.shape
            keep = orc.dropout_keep_ntc(seed, ids[site], b, t, c, p).transpose(0, 2, 1)
        return x * torch.from_numpy(keep.astype(np.float64) / (1.0 - np.float32(p))).to(x.dtype)
    return drop


def test_train_mode_with_dropout_matches_the_oracle():
    """Dropout ON (encoder 0.1, prenet 0.1, decoder 0.05) on a mid-size model with the full configuration's structure
    (mean_only, prenet, 80 mels -> 160 flow channels so that the 80-channel coupling input takes the padded-channel path,
    kernel 5): the oracle (float64, CPU) replays the product's masks site by site; losses 1e-4, gradients 5e-3 rel-L2."""
    enc = dict(n_vocab=30, hidden_channels=64, filter_channels=128, filter_channels_dp=128, kernel_size=3, p_dropout=0.1, n_layers=2,
               n_heads=2, window_size=4, prenet=True, mean_only=True)
    dec = dict(hidden_channels=64, kernel_size=5, n_blocks=3, n_layers=2, n_sqz=2, n_split=4, sigmoid_scale=False, p_dropout=0.05,
               dilation_rate=1)
    cfg = dict(encoder=enc, decoder=dec, zero_out=False)
    p32 = go.init_params(cfg, 30, 80, seed=7)
    model = _build(cfg, 80, p32)
    tokens, x_lens, y, y_lens = go.synthetic_batch(3, 23, 120, 30, 80, seed=8)
    model.train()
    loss_dict, _ = model(tokens.to(DEV), x_lens.to(DEV), y.to(DEV), y_lens.to(DEV))
    loss_dict["loss"].backward()
    seed = model._drop_seed

    def p_of(site):
        return 0.1 if site.startswith("encoder.") else 0.05
    p64 = {k: v.double().requires_grad_(True) for k, v in p32.items()}
    out, aux = go.glow_tts_forward(tokens, x_lens, y.double(), y_lens, p64, cfg, True, drop=_masks_for(model, seed, p_of))
    out["loss"].backward()
    print(f"\n[glow_tts train, dropout on] product {loss_dict['loss_mle'].item():.6f} / {loss_dict['loss_length'].item():.6f}; "
          f"oracle {float(out['loss_mle']):.6f} / {float(out['loss_length']):.6f}")
    assert np.isclose(loss_dict["loss_mle"].item(), float(out["loss_mle"]), rtol=1e-4)
    assert np.isclose(loss_dict["loss_length"].item(), float(out["loss_length"]), rtol=1e-4)
    worst = (0.0, "")
    gmax = max(float(v.grad.norm()) for v in p64.values())
    for name, prm in model.named_parameters():
        ref = p64[name].grad
        # relative L2 per tensor, with an absolute floor: the key bias of an attention layer has an exactly zero gradient
        # (softmax is invariant to a constant added to all keys' scores), which fp32 reproduces as ~1e-9, not as 0
        e = float((prm.grad.double().cpu() - ref).norm() / (ref.norm() + 1e-6 * gmax))
        worst = max(worst, (e, name))
    print(f"  worst gradient rel-L2 {worst}")
    assert worst[0] <= 5e-3, worst


def test_actnorm_data_dependent_init():
    """GlowTTS.ddi (glow_tts.py:49-56): after the data-dependent initialisation every ActNorm output channel of the first
    flow has zero mean and unit variance over the valid frames (ActNorm.initialize, submodules.py:261-274)."""
    cfg = dict(encoder=dict(go.GOLDEN_CFG["encoder"]), decoder=dict(go.GOLDEN_CFG["decoder"]), zero_out=False)
    model = _build(cfg, 8, go.init_params(cfg, 20, 8, seed=3))
    tokens, x_lens, y, y_lens = go.synthetic_batch(4, 12, 60, 20, 8, seed=4)
    batch = [tokens.to(DEV), x_lens.to(DEV), y.to(DEV), y_lens.to(DEV), None, None, None]
    model.ddi(batch)
    an = model.decoder.flows[0]
    with torch.no_grad():
        yl = ((y_lens // 2) * 2).to(DEV)
        x, xl = model.decoder.squeeze(y.to(DEV).transpose(1, 2).contiguous(), yl.to(torch.int32), 2)
        z, _ = an(x.contiguous(), xl)
        keep = (torch.arange(z.shape[1], device=DEV)[None, :] < xl[:, None]).unsqueeze(-1).float()
        n = keep.sum()
        mean = (z * keep).sum((0, 1)) / n
        var = ((z ** 2) * keep).sum((0, 1)) / n - mean ** 2
    assert mean.abs().max() < 1e-4 and (var - 1).abs().max() < 1e-3


def test_train_py_glow_tts_two_epochs(tmp_path, monkeypatch):
    """`train.py --model glow_tts --dataset synthetic_tts` end to end (registry, DataLoader + collate with ragged tokens and
    frames, Noam schedule, AdamW, validation with reconstruction, checkpoint) on a reduced copy of the configuration."""
    import train
    from utils import config as C
    monkeypatch.chdir(PKG)
    m = C.load("configs/models/glow_tts.yaml")
    m.model.encoder.update(C.create(dict(hidden_channels=64, filter_channels=128, n_layers=2)))
    m.model.decoder.update(C.create(dict(hidden_channels=64, n_blocks=3, n_layers=2)))
    m.scheduler.warmup_steps = 50
    C.save(m, "configs/models/_test_glow.yaml")
    ds = C.load("configs/datasets/synthetic_tts.yaml")
    ds.dataset.update(C.create(dict(num_clips=8, max_tokens=24)))
    C.save(ds, "configs/datasets/_test_tts.yaml")
    try:
        log_dir = str(tmp_path / "run")
        train.main(["--model", "_test_glow", "--dataset", "_test_tts", "--batch_size", "4", "--num_workers", "0", "--total_epochs", "2",
                    "--log_every_n_steps", "1", "--eval_every_n_epochs", "1", "--log_dir", log_dir, "--n_gpus", "1"])
        last = torch.load(os.path.join(log_dir, "ckpts", "ckpt.last.pt"), weights_only=True)
        assert last["step"] == 4 and "decoder.flows.2.wn.in_layers.0.weight_g" in last["model"]
        import json
        scal = os.path.join(log_dir, "scalars.jsonl")
        if os.path.exists(scal):
            rows = [json.loads(line) for line in open(scal)]
            assert {"loss/train_loss_mle", "loss/train_loss_length", "loss/val_loss"} <= {r["tag"] for r in rows}
            assert all(np.isfinite(r["value"]) for r in rows)
    finally:
        os.remove("configs/models/_test_glow.yaml")
        os.remove("configs/datasets/_test_tts.yaml")
