"""The CPU oracle against the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import torch

from oracle import vqvae_oracle as orc

T = torch.from_numpy


def small_cfg():
    return orc.VQVAEConfig(width=16, emb_width=32, l_bins=64, multipliers=(1, 1, 1), linf_topk=128)


def params_from(g, prefix="p."):
    return {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}


def test_stft_matches_reference(golden):
    g = golden("stft")
    x = T(g["x"])
    for n_fft, hop, win in [(1024, 256, 1024), (2048, 240, 1200), (1024, 120, 600), (512, 50, 240)]:
        mag = orc.stft_magnitude(x, n_fft, hop, win)
        ref = T(g[f"mag_{n_fft}_{hop}_{win}"])
        assert mag.shape == ref.shape == (2, n_fft // 2 + 1, orc.stft_num_frames(x.shape[-1], n_fft, hop))
        # fp32 tolerance: magnitudes reach ~90; 1e-4 abs is ~1 ulp there
        assert torch.allclose(mag, ref, atol=1e-4, rtol=1e-5)
        basis = orc.stft_forward_basis(n_fft, win)[:, 0]
        rows = [0, 1, n_fft // 4, n_fft // 2, n_fft // 2 + 2]
        assert torch.allclose(basis[rows], T(g[f"basis_rows_{n_fft}_{win}"]), atol=1e-7)


def test_mel_matches_reference_with_restated_filterbank(golden):
    g = golden("mel")
    basis = orc.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert np.array_equal(basis, g["mel_basis"])  # fixture is pinned to the restatement (parity unpinned)
    mel = orc.mel_spectrogram(T(g["x"]), T(basis))
    assert mel.shape == (2, 80, 8192 // 256)
    assert torch.allclose(mel, T(g["mel"]), atol=1e-5)
    # structural properties of a Slaney filterbank
    assert (basis >= 0).all() and (basis.sum(1) > 0).all()
    assert basis[:, -1].sum() == 0  # fmax = 8 kHz < Nyquist


def test_vq_exact_argmin_equals_reference_off_roundoff(golden):
    g = golden("vq_quantize")
    for tag in ("gauss", "enc"):
        idx, d1, d2 = orc.vq_argmin_exact(g[f"{tag}_x"], g[f"{tag}_k"])
        pinned = g[f"{tag}_pinned"]
        assert np.array_equal(idx[pinned], g[f"{tag}_idx"][pinned])
        assert np.allclose(d1, g[f"{tag}_d_best"]) and np.allclose(d2, g[f"{tag}_d_second"])
        ridx, fit, _ = orc.vq_quantize_reference(T(g[f"{tag}_x"]), T(g[f"{tag}_k"]), T(g[f"{tag}_mask"]))
        assert np.array_equal(ridx.numpy(), g[f"{tag}_idx"])
        assert np.isclose(float(fit), float(g[f"{tag}_fit_masked"]), rtol=1e-5)
        _, fit_nm, _ = orc.vq_quantize_reference(T(g[f"{tag}_x"]), T(g[f"{tag}_k"]))
        assert np.isclose(float(fit_nm), float(g[f"{tag}_fit_nomask"]), rtol=1e-5)
    idx, _, _ = orc.vq_argmin_exact(g["tie_x"], g["tie_k"])
    assert np.array_equal(idx, g["tie_idx"]) and idx[1] == 2  # duplicate rows -> lowest index


def test_vq_forward_update_k_matches_reference(golden):
    g = golden("vq_forward")
    mask = T(g["mask"])
    state = orc.CodebookState(k=torch.zeros(48, 32))
    for step in range(3):
        x = T(g[f"s{step}_x"]).requires_grad_(True)
        if step == 0:
            orc.vq_init_k(state, T(g["s0_k_rand_init"]))
        x_l, x_d, commit, metrics = orc.vq_forward(x, mask, state, float(g["mu"]), float(g["threshold"]),
                                                   update_k=True, k_rand=T(g[f"s{step}_k_rand"]))
        (x_d.sum() + commit * 3.0).backward()
        assert np.array_equal(x_l.numpy(), g[f"s{step}_x_l"])
        assert torch.allclose(x_d, T(g[f"s{step}_x_d"]), atol=1e-6)
        assert torch.allclose(commit, T(g[f"s{step}_commit"]), rtol=1e-5)
        assert torch.allclose(x.grad, T(g[f"s{step}_dx"]), atol=1e-6)
        for name in ("k", "k_sum", "k_elem"):
            assert torch.allclose(getattr(state, name), T(g[f"s{step}_{name}"]), atol=1e-6), name
        for mk in ("fit", "entropy", "used_curr", "usage", "dk"):
            assert np.isclose(float(metrics[mk]), float(g[f"s{step}_m_{mk}"]), rtol=1e-5), mk


def test_gated_hifi_block_matches_reference(golden):
    g = golden("gated_hifi")
    p = {"b." + k: v for k, v in params_from(g).items()}
    x = T(g["x"])
    mask = orc.sequence_mask(T(g["lens"]), x.shape[-1]).unsqueeze(1).float()
    cfg = orc.VQVAEConfig(width=16, multipliers=(1, 1, 1))
    y = orc.gated_hifi_block(x, mask, p, "b", cfg, orc.no_dropout)
    assert torch.allclose(y, T(g["y"]), atol=1e-5)


def test_losses_match_reference(golden):
    g = golden("losses")
    y, lens = T(g["y"]), T(g["lens"])
    yh = T(g["yh"]).requires_grad_(True)
    mask = orc.sequence_mask(lens, y.shape[-1]).unsqueeze(1).float()
    cfg = orc.VQVAEConfig()
    ls = orc.multires_stft_loss(y, yh, mask, cfg)
    gs, = torch.autograd.grad(ls, yh)
    assert np.isclose(float(ls.detach()), float(g["loss_stft"]), rtol=1e-5)
    assert torch.allclose(gs, T(g["grad_stft"]), atol=1e-6, rtol=1e-4)
    lr = orc.multinorm_recon_loss(y, yh, mask, cfg)
    gr, = torch.autograd.grad(lr, yh)
    assert np.isclose(float(lr.detach()), float(g["loss_recon"]), rtol=1e-5)
    assert torch.allclose(gr, T(g["grad_recon"]), atol=1e-8, rtol=1e-4)
    cfg_l1 = orc.VQVAEConfig(l1=0.5, linf_topk=64)
    assert np.isclose(float(orc.multinorm_recon_loss(y, yh, mask, cfg_l1)), float(g["loss_recon_l1"]), rtol=1e-5)
    cfg_nl = orc.VQVAEConfig(log_stft=False)
    assert np.isclose(float(orc.multires_stft_loss(y, yh, mask, cfg_nl)), float(g["loss_stft_nolog"]), rtol=1e-5)


def test_small_vqvae_eval_step_and_encode_match_reference(golden):
    g = golden("vqvae_small")
    cfg = small_cfg()
    p = {k: v.requires_grad_(True) for k, v in params_from(g).items()}
    assert set(p) == set(orc.param_shapes(cfg))
    x, lens = T(g["x"]), T(g["lens"])
    mask = orc.sequence_mask(lens, x.shape[-1]).unsqueeze(1).float()
    with torch.no_grad():
        z, zm = orc.encoder_forward(x, mask, p, cfg)
    assert torch.allclose(z, T(g["enc_z"]), atol=1e-5)
    zf, _ = orc.vq_preprocess(z, zm)
    idx, _, _ = orc.vq_argmin_exact(zf.numpy(), g["k0"])
    assert np.array_equal(idx.reshape(3, -1), g["enc_codes"])
    state = orc.CodebookState(k=T(g["k0"]).clone(), init=True)
    out, metrics, _ = orc.vqvae_forward(x, lens, p, cfg, state, training=False)
    assert metrics == {}
    out["loss"].backward()
    for kk in ("loss", "loss_recon", "loss_stft", "loss_commit"):
        assert np.isclose(float(out[kk]), float(g["eval_" + kk]), rtol=2e-5), kk
    assert torch.allclose(out["yh"], T(g["eval_yh"]), atol=1e-5)
    n_checked = 0
    for name, v in g.items():
        if name.startswith("eval_g."):
            grad = p[name[len("eval_g."):]].grad
            assert grad is not None and torch.allclose(grad, T(v), atol=1e-5, rtol=1e-3), name
            n_checked += 1
    assert n_checked > 100


def test_small_vqvae_train_steps_match_reference(golden):
    """Two train-mode steps (dropout p=0, captured k_rand) + AdamW: losses,
    VQ metrics, codebook state and probe gradients (train.py:82-143)."""
    g = golden("vqvae_train")
    cfg = small_cfg()
    p = {k: v.clone().requires_grad_(True) for k, v in params_from(g).items()}
    names = list(orc.param_shapes(cfg))
    opt = torch.optim.AdamW([p[n] for n in names], lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0)
    state = orc.CodebookState(k=torch.zeros(64, 32))
    lens = T(g["lens"])
    for step in range(2):
        x = T(g[f"tr{step}_x"])
        if step == 0:
            orc.vq_init_k(state, T(g["tr0_k_rand_init"]))
        opt.zero_grad()
        out, metrics, _ = orc.vqvae_forward(x, lens, p, cfg, state, training=True,
                                            k_rand=T(g[f"tr{step}_k_rand"]))
        out["loss"].backward()
        for kk in ("loss", "loss_recon", "loss_stft", "loss_commit"):
            assert np.isclose(float(out[kk]), float(g[f"tr{step}_{kk}"]), rtol=5e-5), (step, kk)
        assert torch.allclose(out["yh"], T(g[f"tr{step}_yh"]), atol=2e-5)
        for mk in ("fit", "entropy", "used_curr", "usage", "dk"):
            assert np.isclose(float(metrics[mk]), float(g[f"tr{step}_m_{mk}"]), rtol=1e-4), (step, mk)
        for name in ("k", "k_sum", "k_elem"):
            assert torch.allclose(getattr(state, name), T(g[f"tr{step}_{name}"]), atol=1e-5), name
        for key, v in g.items():
            if key.startswith(f"tr{step}_g."):
                ref = T(v)  # fp32 conv-backward summation order differs run to run: scale by max
                assert (p[key.split("_g.", 1)[1]].grad - ref).abs().max() <= 1e-3 * ref.abs().max(), key
        gnorm = torch.sqrt(sum((p[n].grad ** 2).sum() for n in names))
        assert np.isclose(float(gnorm), float(g[f"tr{step}_gnorm"]), rtol=1e-3)
        opt.step()
        if step == 0:
            # the optimiser step itself: entries with well-determined gradients move identically;
            # AdamW(eps=1e-9) amplifies fp32 round-off on the rest to +-lr, so continue from the
            # reference's own post-step parameters.
            ref1 = params_from(g, "p1.")
            close = sum(int(torch.isclose(p[n].detach(), ref1[n], atol=2e-6).sum()) for n in names)
            total = sum(p[n].numel() for n in names)
            assert close / total > 0.97, close / total
            with torch.no_grad():
                for n in names:
                    p[n].copy_(ref1[n])
    assert torch.allclose(p["decoders.0.out.weight"], T(g["final_w_probe"]), atol=1e-6)


def test_param_ema_matches_reference(golden):
    """EMA.step/swap (models/ema.py:55-66): state = mu*state + (1-mu)*p."""
    g = golden("param_ema")
    mu = 0.9
    sw, sb = T(g["w0"]).clone(), T(g["b0"]).clone()
    for i in range(1, 4):
        sw = sw * mu + (1 - mu) * T(g[f"w{i}"])
        sb = sb * mu + (1 - mu) * T(g[f"b{i}"])
        assert torch.allclose(sw, T(g[f"ema_w{i}"]), atol=1e-6)
        assert torch.allclose(sb, T(g[f"ema_b{i}"]), atol=1e-6)
    assert np.allclose(g["swapped_w"], g["ema_w3"]) and np.allclose(g["swapped_ema_w"], g["w3"])


def test_state_dict_inventory_matches_full_config():
    with open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_inventory.json")) as f:
        inv = json.load(f)
    shapes = orc.param_shapes(orc.VQVAEConfig())
    assert inv["n_trainable"] == 7405441 == sum(int(np.prod(s)) for s in shapes.values())
    buffers = {k for k in inv["entries"] if "basis" in k or k.endswith(".k")}
    assert len(inv["entries"]) == 413 and len(buffers) == 7
    for name, shape in shapes.items():
        assert inv["entries"][name] == list(shape), name
    assert list(shapes) == [k for k in inv["entries"] if k not in buffers]


def test_counter_dropout_statistics():
    keep = orc.dropout_keep_ntc(seed=3, site=5, b=2, t=1000, c=64, p=0.1)
    assert abs(keep.mean() - 0.9) < 0.005
    keep2 = orc.dropout_keep_ntc(seed=3, site=6, b=2, t=1000, c=64, p=0.1)
    assert 0.75 < (keep == keep2).mean() < 0.9  # independent sites: 0.81 + 0.01
    assert len(set(orc.dropout_site_ids(orc.VQVAEConfig()).values())) == 112


def test_mas_oracle_matches_reference_golden(golden):
    """oracle/mas_oracle.py against the path captured from the reference's maximum_path (submodules.py:28-67)."""
    from oracle import mas_oracle
    g = golden("mas")
    path = mas_oracle.maximum_path(g["value"], g["mask"])
    assert np.array_equal(path, g["path"])
    # a monotonic alignment: on the regular items every unmasked frame is assigned to exactly one token, tokens ascend
    for i in (0, 1):
        xl, yl = int(g["x_len"][i]), int(g["y_len"][i])
        p = path[i, :xl, :yl]
        assert (p.sum(0) == 1).all() and (np.diff(p.argmax(0)) >= 0).all() and (np.diff(p.argmax(0)) <= 1).all()


def test_stft_inverse_oracle_matches_reference_golden(golden):
    """oracle.stft_inverse against STFT.inverse of the reference (transforms.py:125-156; window_sumsquare restated, see
    tests/golden/make_golden.py: stft_inverse), and the round trip back to the signal away from the edges."""
    g = golden("stft_inverse")
    for tag in "abc":
        n_fft, hop, win = (int(v) for v in g[f"{tag}_cfg"])
        y = orc.stft_inverse(torch.from_numpy(g[f"{tag}_mag"]), torch.from_numpy(g[f"{tag}_phase"]), n_fft, hop, win)
        assert torch.allclose(y, torch.from_numpy(g[f"{tag}_y"]), atol=1e-6)
        x = torch.from_numpy(g[f"{tag}_x"])
        assert (y[:, 0, n_fft:-n_fft] - x[:, n_fft:y.shape[-1] - n_fft]).abs().max() < 5e-6


def _lm_fixture(golden):
    g = golden("transformer_lm")
    params = {k[len("param."):]: torch.from_numpy(g[k]) for k in g if k.startswith("param.")}
    grads = {k[len("grad."):]: torch.from_numpy(g[k]) for k in g if k.startswith("grad.")}
    return g, params, grads


def test_transformer_lm_oracle_matches_reference_golden(golden):
    """transformer_lm.py:109-135 run by the reference's own class (dropout 0): logits, loss, accuracy and every gradient."""
    from oracle import lm_oracle as lmo
    g, params, grads = _lm_fixture(golden)
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    p = {k: v.double().requires_grad_(True) for k, v in params.items()}
    logits = lmo.lm_logits(x, lens, p, heads=2, num_layers=2)
    loss, acc = lmo.lm_loss(x, logits)
    assert torch.allclose(logits.float(), torch.from_numpy(g["logits"]), atol=1e-5)
    assert abs(loss.item() - float(g["loss"])) < 1e-6 and abs(acc.item() - float(g["accuracy"])) < 1e-7
    loss.backward()
    for name, ref in grads.items():
        if name.startswith("vqvae."):
            continue
        mine = p[name].grad.float()
        assert torch.allclose(mine, ref, atol=2e-6, rtol=1e-4), (name, float((mine - ref).abs().max()))
    assert float(grads["embedding.weight"][lmo.PAD].abs().max()) == 0.0       # padding_idx row: no gradient


def test_transformer_lm_oracle_dropout_sites_are_unbiased():
    from oracle import lm_oracle as lmo
    drop = lmo.CounterDropout(seed=3, p=0.1)
    x = torch.ones(4, 2, 33, 33)
    a, b = drop(1, x), drop(5, x)
    assert abs(float(a.mean()) - 1.0) < 0.02 and set(a.unique().tolist()) == {0.0, float(np.float32(1) / np.float32(0.9))}
    assert not torch.equal(a, b)                                    # sites draw different masks
    assert torch.equal(lmo.CounterDropout(seed=3, p=0.0)(1, x), x)


def test_glow_tts_oracle_matches_reference_golden(golden):
    """oracle/glow_oracle.py (float64) against the reference's own GlowTTS (tests/golden/glow_tts.npz): losses, the
    alignment found by the monotonic search, the latent, the log-determinants and every parameter gradient."""
    import torch
    from oracle import glow_oracle as go
    g = golden("glow_tts")
    params = {k[len("param."):]: torch.from_numpy(g[k]).double().requires_grad_(True) for k in g if k.startswith("param.")}
    cfg = go.GOLDEN_CFG
    out, aux = go.glow_tts_forward(torch.from_numpy(g["tokens"]), torch.from_numpy(g["x_lens"]), torch.from_numpy(g["y"]).double(),
                                   torch.from_numpy(g["y_lens"]), params, cfg, True)
    assert np.isclose(float(out["loss_mle"]), float(g["loss_mle"]), rtol=1e-5) and np.isclose(float(out["loss_length"]), float(g["loss_length"]), rtol=1e-5)
    assert np.array_equal(aux["attn"].numpy().astype(np.float32), g["attn"])
    assert np.allclose(aux["z_dec"].detach().numpy(), g["z_dec"], atol=1e-5) and np.allclose(aux["logdet"].detach().numpy(), g["logdet"], rtol=1e-5)
    out["loss"].backward()
    checked = 0
    for k in g:
        if k.startswith("grad."):
            ref, got = g[k], params[k[len("grad."):]].grad
            assert got is not None and np.linalg.norm(got.numpy() - ref) <= 1e-4 * np.linalg.norm(ref) + 1e-7, k
            checked += 1
    assert checked > 80
    ev, _ = go.glow_tts_forward(torch.from_numpy(g["tokens"]), torch.from_numpy(g["x_lens"]), torch.from_numpy(g["y"]).double(),
                                torch.from_numpy(g["y_lens"]), {k: v.detach() for k, v in params.items()}, cfg, False,
                                noise=torch.from_numpy(g["eval_noise"]).double())
    assert np.allclose(ev["yh"].numpy(), g["eval_yh"], atol=1e-4)
