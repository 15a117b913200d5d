#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE
(vliu15/speech-masters-thesis at /root/reference) on CPU in the build
container.  The reference never travels to the GPU box -- only the small .npz
fixtures written here do.  Run from the repo root:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Harness notes:
  * /root/reference goes first on sys.path (a pip package named `datasets`
    would otherwise shadow the reference's `datasets/`), cwd = /root/reference
    so `logger.conf` resolves;
  * librosa is absent offline: an in-memory module provides
    `librosa.util.pad_center` / `tiny` (the only librosa calls on the STFT
    path, transforms.py:97-98) and `librosa.filters.mel` is routed to the
    oracle's Slaney restatement -- the mel fixture is therefore labelled
    "restated filterbank" (parity unpinned for librosa.filters.mel itself);
  * OmegaConf is absent: a dict with attribute access stands in.
Everything numeric in the fixtures is produced by the reference's own classes.
"""
import json
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
os.chdir(REF)
sys.dont_write_bytecode = True

import logging.config  # noqa: E402  (models/ema.py uses logging.config after a bare import)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import yaml  # noqa: E402

from oracle import vqvae_oracle as orc  # noqa: E402

# ---- librosa stand-in --------------------------------------------------------
librosa = types.ModuleType("librosa")
librosa.util = types.ModuleType("librosa.util")
librosa.filters = types.ModuleType("librosa.filters")


def _pad_center(data, size):
    n = data.shape[-1]
    lpad = (size - n) // 2
    out = np.zeros(size, dtype=data.dtype)
    out[lpad:lpad + n] = data
    return out


librosa.util.pad_center = _pad_center
librosa.util.tiny = lambda x: np.finfo(np.float32).tiny
librosa.filters.mel = lambda sr, n_fft, n_mels, fmin, fmax: orc.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
sys.modules["librosa"] = librosa
sys.modules["librosa.util"] = librosa.util
sys.modules["librosa.filters"] = librosa.filters

from datasets.transforms import STFT, MelSpectrogram  # noqa: E402
from models.vqvae.bottleneck import BottleneckBlock  # noqa: E402
from models.vqvae.losses import MultiNormReconstructionLoss, MultiResolutionSpectralLoss  # noqa: E402
from models.vqvae.resnet import GatedHiFiBlock  # noqa: E402
from models.vqvae.vqvae import VQVAE  # noqa: E402
from models.ema import EMA  # noqa: E402


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def wrap(o):
    if isinstance(o, dict):
        return AttrDict({k: wrap(v) for k, v in o.items()})
    return o


def load_cfg(**model_overrides):
    with open(os.path.join(REF, "configs/models/vqvae.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model"].update(model_overrides)
    return wrap(cfg)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def synth(b, t, seed):
    return orc.synthetic_clip_batch(b, t, seed)[:, 0]


# ---- G1: STFT magnitudes (transforms.py:108-123), 4 parameter sets -------------
def gen_stft():
    x = synth(2, 4608, 11)
    out = {"x": x}
    for n_fft, hop, win in [(1024, 256, 1024), (2048, 240, 1200), (1024, 120, 600), (512, 50, 240)]:
        mod = STFT(n_fft=n_fft, hop_length=hop, win_length=win, window="hann")
        mag = mod(x)
        out[f"mag_{n_fft}_{hop}_{win}"] = mag
        mine = orc.stft_magnitude(x, n_fft, hop, win)
        err = (mine - mag).abs().max().item()
        print(f"  stft {n_fft}/{hop}/{win}: oracle vs ref max abs {err:.3e} (max {mag.max():.2f})")
        assert err < 2e-4
        # basis corner samples pin the window/basis restatement
        out[f"basis_rows_{n_fft}_{win}"] = mod.forward_basis[[0, 1, n_fft // 4, n_fft // 2, n_fft // 2 + 2], 0, :]
    save("stft", **out)


# ---- G2: log-mel (transforms.py:48-65) with the RESTATED Slaney filterbank -----
def gen_mel():
    x = synth(2, 8192, 12)
    mod = MelSpectrogram(sample_rate=22050, n_fft=1024, win_length=1024, hop_length=256, n_mels=80,
                         f_min=0.0, f_max=8000.0)
    mel = mod(x)
    mine = orc.mel_spectrogram(x, mod.mel_basis)
    print("  mel oracle vs ref max abs", (mine - mel).abs().max().item())
    save("mel", x=x, mel=mel, mel_basis=mod.mel_basis)


# ---- G3: vector quantiser (bottleneck.py) ---------------------------------------
def margins(x, k):
    idx, d1, d2 = orc.vq_argmin_exact(x.numpy(), k.numpy())
    return idx, d1, d2


def gen_vq():
    out = {}
    g = torch.Generator().manual_seed(21)
    for tag, (n, d, kb) in {"gauss": (768, 128, 256), "enc": (640, 128, 1024)}.items():
        x = torch.randn(n, d, generator=g)
        if tag == "gauss":
            k = torch.randn(kb, d, generator=g)
        else:  # encoder-like: codebook drawn from (tiled, jittered) data rows
            x = x * 0.3 + torch.randn(1, d, generator=g)
            rows = x.repeat(2, 1) + torch.randn(2 * n, d, generator=g) * 0.02
            k = rows[torch.randperm(2 * n, generator=g)][:kb]
        blk = BottleneckBlock(kb, d, 0.99, 1.0)
        blk.k = k.clone()
        mask = torch.ones(n, 1)
        mask[n - 37:] = 0
        x_l, fit = blk.quantize(x, mask)
        x_l_nomask, fit_nomask = blk.quantize(x)
        assert torch.equal(x_l, x_l_nomask)
        idx, d1, d2 = margins(x, k)
        # Margin relative to the NORMS (the fp32 expression sum(x^2) - 2xk + sum(k^2) loses
        # ~1e-7 * (|x|^2 + |k|^2) to cancellation, bottleneck.py:129-133).  Rows under 1e-5
        # are at fp32 round-off: there the reference's own answer depends on the sgemm
        # summation order and is not a pin; everywhere else it must equal the exact argmin.
        scale = (x.numpy().astype(np.float64) ** 2).sum(1) + (k.numpy().astype(np.float64) ** 2).sum(1).max()
        nrel = (d2 - d1) / scale
        pinned = nrel > 1e-5
        agree = (idx == x_l.numpy())
        print(f"  vq[{tag}] ref-vs-exact agreement {agree.mean():.6f}; pinned rows {pinned.mean():.4f}; "
              f"min norm-rel margin {nrel.min():.3e}")
        assert agree[pinned].all(), "reference must equal the exact argmin off round-off margins"
        out[f"{tag}_pinned"] = pinned
        out.update({f"{tag}_x": x, f"{tag}_k": k, f"{tag}_mask": mask, f"{tag}_idx": x_l,
                    f"{tag}_fit_masked": fit, f"{tag}_fit_nomask": fit_nomask,
                    f"{tag}_d_best": d1, f"{tag}_d_second": d2})
    # tie rule: duplicated codebook rows -> lowest index (torch.min, bottleneck.py:134)
    k = torch.randn(8, 16, generator=g)
    k[5] = k[2]
    x = k[[2, 5, 7]] + 0.0
    blk = BottleneckBlock(8, 16, 0.99, 1.0)
    blk.k = k
    x_l, _ = blk.quantize(x)
    out.update(tie_x=x, tie_k=k, tie_idx=x_l)
    save("vq_quantize", **out)


def gen_vq_forward():
    """BottleneckBlock.forward(update_k=True) twice (init_k then a regular
    update) with torch.randperm / randn_like captured, so `_k_rand` is a
    fixture INPUT (bottleneck.py:35-46, 60-90, 171-201)."""
    g = torch.Generator().manual_seed(31)
    b, d, t, kb = 3, 32, 40, 48
    lens = torch.tensor([40, 33, 17])
    mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
    blk = BottleneckBlock(kb, d, 0.9, 1.0)
    blk.train()
    captured = []
    real_randperm = torch.randperm

    def fake_randperm(n, *a, **kw):
        p = real_randperm(n, generator=g)
        captured.append(p)
        return p

    torch.randperm = fake_randperm
    try:
        out = {"mask": mask, "lens": lens, "mu": 0.9, "threshold": 1.0}
        for step in range(3):
            x = torch.randn(b, d, t, generator=g) * (1.0 + step)
            x.requires_grad_(True)
            x_l, x_d, commit, metrics = blk(x, mask, update_k=True)
            (x_d.sum() + commit * 3.0).backward()
            xf, mf = orc.vq_preprocess(x.detach(), mask)
            rows = xf[(mf != 0)[:, 0]]
            # reconstruct the k_rand rows the reference drew (rows >= K here so _tile is identity)
            assert rows.shape[0] >= kb
            perms = captured[-(2 if step == 0 else 1):]
            out[f"s{step}_x"] = x.detach()
            if step == 0:
                out["s0_k_rand_init"] = rows[perms[0]][:kb]
            out[f"s{step}_k_rand"] = rows[perms[-1]][:kb]
            out[f"s{step}_x_l"] = x_l
            out[f"s{step}_x_d"] = x_d.detach()
            out[f"s{step}_commit"] = commit.detach()
            out[f"s{step}_dx"] = x.grad
            for mk, mv in metrics.items():
                out[f"s{step}_m_{mk}"] = mv
            out[f"s{step}_k"] = blk.k.clone()
            out[f"s{step}_k_sum"] = blk.k_sum.clone()
            out[f"s{step}_k_elem"] = blk.k_elem.clone()
            k_used = out["s0_k_rand_init"] if step == 0 else out[f"s{step - 1}_k"]
            idx, d1, d2 = orc.vq_argmin_exact(xf.numpy(), k_used.numpy())
            assert (idx == x_l.reshape(-1).numpy()).all()
            out[f"s{step}_min_rel_margin"] = ((d2 - d1) / np.maximum(d1, 1e-30)).min()
    finally:
        torch.randperm = real_randperm
    save("vq_forward", **out)


# ---- G4: one GatedHiFiBlock in eval mode (resnet.py:184-241) ---------------------
def randomize(module, g, scale=1.0):
    with torch.no_grad():
        for p in module.parameters():
            if p.abs().sum() == 0:  # zero_out layers: give them signal
                fan_in = p.shape[1] * p.shape[2] if p.dim() == 3 else 16
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * scale / np.sqrt(fan_in))


def gen_block():
    g = torch.Generator().manual_seed(41)
    torch.manual_seed(41)           # the module constructor draws from the GLOBAL generator: seed it, so the fixture regenerates
    blk = GatedHiFiBlock(16, 4, dilation_growth_rate=3, kernel_size_growth_rate=2, zero_out=True)
    randomize(blk, g)
    blk.eval()
    b, t = 2, 300
    x = torch.randn(b, 16, t, generator=g)
    lens = torch.tensor([300, 211])
    mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
    y, _ = blk(x, mask)
    cfg = orc.VQVAEConfig(width=16, multipliers=(1, 1, 1))
    p = {"b." + k: v for k, v in blk.state_dict().items()}
    mine = orc.gated_hifi_block(x, mask, p, "b", cfg, orc.no_dropout)
    print("  gated_hifi oracle vs ref", (mine - y).abs().max().item())
    assert torch.allclose(mine, y, atol=1e-5)
    save("gated_hifi", x=x, lens=lens, y=y, **{"p." + k: v for k, v in blk.state_dict().items()})


# ---- G5/G7: small VQVAE, eval + deterministic train steps -------------------------
SMALL = dict(width=16, emb_width=32, l_bins=64, multipliers=[1, 1, 1])


def small_loss_cfg(cfg):
    cfg.model.loss.linf_topk = 128
    return cfg


def gen_model():
    g = torch.Generator().manual_seed(51)
    cfg = small_loss_cfg(load_cfg(**SMALL))
    torch.manual_seed(5)
    model = VQVAE(cfg)
    randomize(model, g, scale=0.5)
    sd = {k: v.clone() for k, v in model.state_dict().items() if "basis" not in k}
    b, t = 3, 2048
    x = synth(b, t, 52).unsqueeze(1)
    lens = torch.tensor([2048, 1536, 1024])
    out = {"x": x, "lens": lens}
    out.update({"p." + k: v for k, v in sd.items() if k != "bottleneck.level_blocks.0.k"})

    # (a) encode-only with a fixed codebook (generate_vq_dataset.py:61-70)
    model.eval()
    mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
    with torch.no_grad():
        z, zm = model.encoders[0](x, mask)
    zf, mf = orc.vq_preprocess(z, zm)
    k0 = zf[(mf != 0)[:, 0]][torch.randperm(int(mf.sum()), generator=g)][:64].clone()
    model.bottleneck.level_blocks[0].k = k0.clone()
    with torch.no_grad():
        codes = model.bottleneck.level_blocks[0].encode(z, zm)
    out.update(enc_z=z, k0=k0, enc_codes=codes)

    # (b) eval-mode supervised_step + backward (deterministic; vqvae.py:98-132)
    batch = [None, None, None, None, x, lens, None]
    model.zero_grad()
    loss_dict, metrics = model.supervised_step(batch)
    assert metrics == {}
    loss_dict["loss"].backward()
    for kk in ("loss", "loss_recon", "loss_stft", "loss_commit", "yh"):
        out["eval_" + kk] = loss_dict[kk].detach()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    out.update({"eval_g." + n: v for n, v in grads.items()})
    # eval: x_quantized is detached (bottleneck.py:230-233) so encoder grads only come from commit
    assert any(n.startswith("encoders") for n in grads)

    save("vqvae_small", **out)

    # state-dict key/shape inventory of the FULL default config
    full = VQVAE(load_cfg())
    inv = {k: list(v.shape) for k, v in full.state_dict().items()}
    n_train = sum(p.numel() for p in full.parameters() if p.requires_grad)
    with open(os.path.join(OUT, "state_dict_inventory.json"), "w") as f:
        json.dump({"n_trainable": n_train, "entries": inv}, f, indent=0)
    print(f"  full config: {len(inv)} state_dict entries, {n_train} trainable")


def gen_model_train_krand():
    """Train-mode forward with rows >= K so `_k_rand` is reconstructible from
    the captured permutation (T = 8192 -> 64+48+32 = 144 latent rows, K = 64)."""
    g = torch.Generator().manual_seed(71)
    cfg = small_loss_cfg(load_cfg(**SMALL))
    torch.manual_seed(7)
    model = VQVAE(cfg)
    randomize(model, g, scale=0.5)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    b, t = 3, 8192
    lens = torch.tensor([8192, 6144, 4096])
    mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
    out = {"lens": lens}
    out.update({"p." + k: v.clone() for k, v in model.state_dict().items()
                if "basis" not in k and k != "bottleneck.level_blocks.0.k"})
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0)
    captured = []
    real_randperm = torch.randperm

    def fake_randperm(n, *a, **kw):
        p = real_randperm(n, generator=g)
        captured.append(p)
        return p

    torch.randperm = fake_randperm
    try:
        for step in range(2):
            captured.clear()
            xs = synth(b, t, 80 + step).unsqueeze(1)
            with torch.no_grad():
                z, zm = model.encoders[0](xs, mask)
            zf, mf = orc.vq_preprocess(z, zm)
            rows = zf[(mf != 0)[:, 0]]
            assert rows.shape[0] >= 64
            opt.zero_grad()
            loss_dict, metrics = model.supervised_step([None, None, None, None, xs, lens, None])
            loss_dict["loss"].backward()
            grads = {n: p.grad.clone() for n, p in model.named_parameters()}
            opt.step()
            if step == 0:
                # AdamW with eps=1e-9 turns fp32 round-off on weakly determined gradients into
                # +-lr parameter differences, so step 1 is pinned from the reference's own
                # post-step-0 parameters rather than from a re-run of the optimiser.
                out.update({"p1." + k: v.detach().clone() for k, v in model.named_parameters()})
                assert len(captured) == 2
                out["tr0_k_rand_init"] = rows[captured[0]][:64]
            out[f"tr{step}_k_rand"] = rows[captured[-1]][:64]
            out[f"tr{step}_x"] = xs
            for kk in ("loss", "loss_recon", "loss_stft", "loss_commit", "yh"):
                out[f"tr{step}_{kk}"] = loss_dict[kk].detach()
            for mk, mv in metrics.items():
                out[f"tr{step}_m_{mk}"] = mv.detach()
            blk = model.bottleneck.level_blocks[0]
            out[f"tr{step}_k"] = blk.k.clone()
            out[f"tr{step}_k_sum"] = blk.k_sum.clone()
            out[f"tr{step}_k_elem"] = blk.k_elem.clone()
            for probe in ("decoders.0.out.weight", "encoders.0.level_blocks.0.blocks.0.weight",
                          "decoders.0.level_blocks.2.blocks.1.gate.weight"):
                out[f"tr{step}_g.{probe}"] = grads[probe]
            out[f"tr{step}_gnorm"] = torch.sqrt(sum((v ** 2).sum() for v in grads.values()))
        out["final_w_probe"] = model.decoders[0].out.weight.detach().clone()
    finally:
        torch.randperm = real_randperm
    save("vqvae_train", **out)


# ---- G6: losses (losses.py) --------------------------------------------------------
def gen_losses():
    g = torch.Generator().manual_seed(61)
    b, t = 3, 6144
    y = synth(b, t, 62).unsqueeze(1)
    yh = (y + 0.05 * torch.randn(b, 1, t, generator=g)).clamp(-1, 1)
    yh.requires_grad_(True)
    lens = torch.tensor([6144, 5120, 3072])
    mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
    stft_loss = MultiResolutionSpectralLoss(n_ffts=[2048, 1024, 512], hop_lengths=[240, 120, 50],
                                            win_lengths=[1200, 600, 240], window="hann", log=True)
    recon = MultiNormReconstructionLoss(l1=0.0, l2=1.0, linf=0.02, linf_topk=2048)
    ls = stft_loss(y, yh, mask)
    g_stft, = torch.autograd.grad(ls, yh)
    lr = recon(y, yh, mask)
    g_recon, = torch.autograd.grad(lr, yh)
    recon_l1 = MultiNormReconstructionLoss(l1=0.5, l2=1.0, linf=0.02, linf_topk=64)
    lr1 = recon_l1(y, yh, mask)
    stft_nolog = MultiResolutionSpectralLoss(n_ffts=[2048, 1024, 512], hop_lengths=[240, 120, 50],
                                             win_lengths=[1200, 600, 240], window="hann", log=False)
    ls_nolog = stft_nolog(y, yh, mask)
    save("losses", y=y, yh=yh.detach(), lens=lens, loss_stft=ls.detach(), grad_stft=g_stft,
         loss_recon=lr.detach(), grad_recon=g_recon, loss_recon_l1=lr1.detach(), loss_stft_nolog=ls_nolog.detach())


# ---- G8: parameter EMA (models/ema.py:24-66) -----------------------------------------
def gen_ema():
    g = torch.Generator().manual_seed(91)
    torch.manual_seed(91)           # nn.Linear's init draws from the global generator
    lin = torch.nn.Linear(5, 3)
    ema = EMA(lin, mu=0.9)
    out = {"w0": lin.weight.detach().clone(), "b0": lin.bias.detach().clone()}
    real_add_ = torch.Tensor.add_

    # `state.mul_(mu).add_(1 - mu, p)` is the removed legacy overload add_(Number, Tensor)
    def compat_add_(self, *a, **kw):
        if len(a) == 2 and not isinstance(a[0], torch.Tensor):
            return real_add_(self, a[1], alpha=a[0])
        return real_add_(self, *a, **kw)

    torch.Tensor.add_ = compat_add_
    try:
        for i in range(3):
            with torch.no_grad():
                lin.weight.add_(torch.randn(3, 5, generator=g))
                lin.bias.add_(torch.randn(3, generator=g))
            ema.step()
            out[f"w{i + 1}"] = lin.weight.detach().clone()
            out[f"b{i + 1}"] = lin.bias.detach().clone()
            out[f"ema_w{i + 1}"] = ema.state_dict()["weight"].clone()
            out[f"ema_b{i + 1}"] = ema.state_dict()["bias"].clone()
        ema.swap()
        out["swapped_w"] = lin.weight.detach().clone()
        out["swapped_ema_w"] = ema.state_dict()["weight"].clone()
    finally:
        torch.Tensor.add_ = real_add_
    save("param_ema", **out)


def gen_stft_inverse():
    """STFT.inverse (datasets/transforms.py:125-156) run by the reference itself.  It calls librosa.filters.window_sumsquare
    (librosa is absent): that ONE helper is supplied by the oracle's restatement of its documented algorithm -- the
    fixture pins the reference's inverse basis, conv_transpose1d, normalisation rule, scaling and trimming; the sum-square
    helper itself stays parity-unpinned (stated in DESIGN.md)."""
    librosa.filters.window_sumsquare = lambda window, n_frames, hop_length, win_length, n_fft, dtype=np.float32: \
        orc.window_sumsquare(n_fft, win_length, hop_length, n_frames)
    g = torch.Generator().manual_seed(111)
    out = {}
    for tag, (n_fft, hop, win) in {"a": (1024, 256, 1024), "b": (512, 50, 240), "c": (2048, 240, 1200)}.items():
        stft = STFT(n_fft=n_fft, hop_length=hop, win_length=win, window="hann")
        x = synth(1, 3 * n_fft + 1000, 7 + n_fft)
        frames = stft(x).shape[-1]
        # a consistent (magnitude, phase) pair of the signal itself, from the reference's own forward bases
        pad = (n_fft - hop) // 2
        xp = F.pad(x.unsqueeze(1), (pad, pad), mode="reflect")
        ft = F.conv1d(xp, stft.forward_basis, stride=hop)
        cutoff = n_fft // 2 + 1
        re, im = ft[:, :cutoff], ft[:, cutoff:]
        mag, phase = torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)
        assert mag.shape[-1] == frames
        y = stft.inverse(mag, phase)
        mine = orc.stft_inverse(mag, phase, n_fft, hop, win)
        print(f"  stft_inverse {tag}: oracle vs ref {float((mine - y).abs().max()):.2e}; round trip vs x "
              f"{float((y[:, 0, n_fft:-n_fft] - x[:, n_fft:y.shape[-1] - n_fft]).abs().max()):.2e}")
        assert torch.allclose(mine, y, atol=1e-6)
        out.update({f"{tag}_mag": mag, f"{tag}_phase": phase, f"{tag}_y": y, f"{tag}_x": x,
                    f"{tag}_cfg": np.array([n_fft, hop, win])})
    save("stft_inverse", **out)


def gen_mas():
    """models/glow_tts/submodules.py:28-67 `maximum_path`, the reference's own function.  It spells numpy's bool as
    `np.bool`, an alias numpy >= 1.24 no longer has: restored in memory for the call (same meaning, nothing else touched)."""
    from oracle import mas_oracle
    if not hasattr(np, "bool"):
        np.bool = bool
    from models.glow_tts.submodules import maximum_path
    g = torch.Generator().manual_seed(101)
    b, t_x, t_y = 4, 37, 90
    value = torch.randn(b, t_x, t_y, generator=g) * 3.0
    x_len = torch.tensor([37, 20, 5, 1])
    y_len = torch.tensor([90, 64, 3, 90])                  # item 2: fewer frames than tokens (the index wraps in numpy)
    mask = ((torch.arange(t_x)[None, :, None] < x_len[:, None, None]) &
            (torch.arange(t_y)[None, None, :] < y_len[:, None, None])).float()
    path = maximum_path(value, mask)
    mine = mas_oracle.maximum_path(value.numpy(), mask.numpy())
    print("  mas oracle vs ref equal:", np.array_equal(mine, path.numpy()))
    assert np.array_equal(mine, path.numpy())
    save("mas", value=value, mask=mask, path=path, x_len=x_len, y_len=y_len)


def gen_transformer_lm():
    """models/transformer_lm/transformer_lm.py:32-135, the reference's own TransformerLM on a small config (2 layers,
    d_model 64, 2 heads of 32, feed-forward 128, vocab 16), dropout 0 (its dropout is torch's global RNG), ragged lengths.
    The module imports OmegaConf for `load_vqvae` only: omegaconf is absent, so an empty module carrying the two NAMES lets
    the import line resolve, and `load_vqvae` (disk access, the VQ-VAE is not on the scored path) is replaced by an empty
    ModuleDict -- nothing of omegaconf is called.  Captured: logits (classifier output), loss, accuracy, all gradients."""
    from oracle import lm_oracle
    om = types.ModuleType("omegaconf")
    om.OmegaConf = type("OmegaConf", (), {})
    om.DictConfig = dict
    sys.modules.setdefault("omegaconf", om)
    from models.transformer_lm.transformer_lm import TransformerLM
    TransformerLM.load_vqvae = staticmethod(lambda log_dir, ckpt_num: torch.nn.ModuleDict())
    cfg = wrap({"model": dict(vocab_size=16, embed_dim=64, max_len=64, num_layers=2, d_model=64, nhead=2, dim_feedforward=128,
                              dropout=0.0, activation="relu", layer_norm_eps=1e-5, norm_first=False, loss_type="ce",
                              vqvae=dict(log_dir="", ckpt_num=0))})
    torch.manual_seed(111)
    model = TransformerLM(cfg)
    # nn.TransformerEncoder clones one layer: perturb so that the two layers (and the LayerNorm affine maps) differ
    g = torch.Generator().manual_seed(112)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if name != "embedding.weight":
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
    model.train()
    x, lens = lm_oracle.synthetic_tokens(3, 12, 16, seed=116)
    x[2, 5] = lm_oracle.BOS          # a special token inside a sequence: not scored (loss_mask, :124)
    captured = {}
    hook = model.classifier.register_forward_hook(lambda mod, inp, out: captured.__setitem__("xh", out.detach()))
    loss_dict, metrics = model(x, lens, None, None)
    hook.remove()
    loss_dict["loss"].backward()
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if k != "pos_encoding.pe"}
    grads = {"grad." + k: v.grad.detach().clone() for k, v in model.named_parameters()}
    logits = captured["xh"].permute(1, 0, 2).contiguous()              # [L, B, V] -> [B, L, V]
    # the restatement against the reference
    p64 = {k: v.double() for k, v in params.items()}
    mine = lm_oracle.lm_logits(x, lens, p64, heads=2, num_layers=2)
    ml, ma = lm_oracle.lm_loss(x, mine)
    print("  oracle vs ref: logits", float((mine.float() - logits).abs().max()), "loss", float(ml) - loss_dict["loss"].item(),
          "acc", float(ma), float(metrics["accuracy"]))
    assert torch.allclose(mine.float(), logits, atol=2e-5) and abs(float(ml) - loss_dict["loss"].item()) < 1e-5
    assert torch.equal(model.pos_encoding.pe[:, 0], lm_oracle.positional_table(64, 64))
    save("transformer_lm", x=x, lens=lens, logits=logits, loss=loss_dict["loss"].detach(), accuracy=metrics["accuracy"],
         **{"param." + k: v for k, v in params.items()}, **grads)


def gen_glow_tts():
    """models/glow_tts/glow_tts.py:12-130, the reference's own GlowTTS (TextEncoder + FlowSpecDecoder + numpy maximum_path +
    MLE / duration losses) on a small configuration: hidden 64, 2 heads, 2 encoder layers, relative window 4, prenet, WN kernel 3 with dilation rate 2,
    mean_only false (so that proj_s / x_logs are exercised); 2 flow blocks of 2 WN layers over 8 mels x n_sqz 2, n_split 4;
    every dropout 0 (the reference draws from torch's global RNG), ragged token and frame lengths.  `models.parser` needs
    inflect / unidecode and a CMUdict file, none of which exist here, and none of which is on the scored path: a name-only
    in-memory module provides `CMUDictParser` and the model is fed token ids directly.  The `end` convolutions of the coupling
    blocks and the prenet's `proj` start at zero in the reference (the flows would be identities): they are perturbed, like
    every other parameter, so that all gradients are informative.  Captured: losses, alignment, z_dec, logdet, every gradient,
    and an eval-mode reconstruction with the noise tensor captured."""
    from oracle import glow_oracle as go
    if not hasattr(np, "bool"):
        np.bool = bool
    parser = types.ModuleType("models.parser")
    parser.CMUDictParser = lambda path: None
    sys.modules.setdefault("models.parser", parser)
    from models.glow_tts.glow_tts import GlowTTS
    enc = dict(n_vocab=20, hidden_channels=64, filter_channels=128, filter_channels_dp=64, kernel_size=3, p_dropout=0.0,
               n_layers=2, n_heads=2, window_size=4, prenet=True, mean_only=False)
    dec = dict(hidden_channels=64, kernel_size=3, n_blocks=2, n_layers=2, n_sqz=2, n_split=4, sigmoid_scale=False, p_dropout=0.0,
               dilation_rate=2)
    cfg = wrap({"model": dict(n_speakers=1, gin_channels=0, encoder=enc, decoder=dec),
                "dataset": dict(n_mels=8, intersperse_blanks=False, cmudict_path="")})
    torch.manual_seed(131)
    model = GlowTTS(cfg)
    g = torch.Generator().manual_seed(132)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.add_(torch.randn(prm.shape, generator=g) * (0.05 if ".end." in name or ".proj." in name else 0.02))
    # the prenet's relu_drop is nn.Dropout(0.1) whatever p_dropout says (modules.py:62): switch it off for a deterministic fixture
    model.encoder.pre.relu_drop[1].p = 0.0
    tokens, x_lens, y, y_lens = go.synthetic_batch(3, 11, 46, 20, 8, seed=133)
    model.train()
    loss_dict, _ = model(tokens, x_lens, y, y_lens)
    loss_dict["loss"].backward()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {"grad." + k: v.grad.detach().clone() for k, v in model.named_parameters()}
    # the restatement against the reference (float64 restatement vs float32 reference)
    p64 = {k: v.double() for k, v in params.items()}
    ocfg = dict(encoder=enc, decoder=dec)
    out, aux = go.glow_tts_forward(tokens, x_lens, y.double(), y_lens, p64, ocfg, True)
    print("  oracle vs ref: loss_mle", float(out["loss_mle"]) - loss_dict["loss_mle"].item(), "loss_length",
          float(out["loss_length"]) - loss_dict["loss_length"].item())
    assert abs(float(out["loss_mle"]) - loss_dict["loss_mle"].item()) < 1e-5
    assert abs(float(out["loss_length"]) - loss_dict["loss_length"].item()) < 1e-5
    # eval mode with the noise captured (glow_tts.py:103-112 draws torch.randn_like)
    model.eval()
    with torch.no_grad():
        torch.manual_seed(134)
        ev, _ = model(tokens, x_lens, y, y_lens)
        torch.manual_seed(134)
        noise = torch.randn(3, 8, 46)
        oe, _ = go.glow_tts_forward(tokens, x_lens, y.double(), y_lens, p64, ocfg, False, noise=noise.double())
    print("  eval yh oracle vs ref", float((oe["yh"].float() - ev["yh"]).abs().max()), tuple(ev["yh"].shape))
    assert torch.allclose(oe["yh"].float(), ev["yh"], atol=1e-4)
    save("glow_tts", tokens=tokens, x_lens=x_lens, y=y, y_lens=y_lens, loss_mle=loss_dict["loss_mle"].detach(),
         loss_length=loss_dict["loss_length"].detach(), attn=aux["attn"].float(), z_dec=aux["z_dec"].float(),
         logdet=aux["logdet"].float(), eval_yh=ev["yh"], eval_noise=noise,
         **{"param." + k: v for k, v in params.items()}, **grads)


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    for name, fn in [("stft", gen_stft), ("mel", gen_mel), ("vq", gen_vq), ("vq_forward", gen_vq_forward),
                     ("block", gen_block), ("losses", gen_losses), ("model", gen_model),
                     ("model_train", gen_model_train_krand), ("ema", gen_ema), ("mas", gen_mas), ("stft_inverse", gen_stft_inverse),
                     ("transformer_lm", gen_transformer_lm), ("glow_tts", gen_glow_tts)]:
        if only and name not in only:
            continue
        print(f"[{name}]")
        fn()
