"""Parity of the EXACT code path bench.py runs -- width 64, bf16, the fused / weight-stationary kernels --
against the oracle, as a composition (VERDICT r01 "What's weak" #1, #2).

  (a) one GatedHiFiBlock(64, 4), train mode with counter dropout, ragged lens, 72,000 rows, so that every fast
      kernel is dispatched (asserted by name): forward and every gradient vs oracle.gated_hifi_block.
  (b) the whole vqvae_k1024 model (bf16 conv stacks), eval mode, two full-length clips: encoder output, code
      indices (bit-exact vs the exact argmin on the model's own z), decoder output and the four losses.
  (c) one full-size TRAIN step (B = 1, 145,408 samples, counter dropout, injected revival rows): losses,
      VQ metrics, codebook state, and the global gradient.
  (d) the small reference fixture (tests/golden/vqvae_small.npz, captured from the reference) through the bf16 path.

bf16 tolerances (stated once, used below; DESIGN.md section 4 quotes them): operands are rounded to bf16
(2^-9 relative) before every GEMM and accumulated in fp32; activations are stored in bf16 between kernels.
Relative L2 error of a tensor after a block: <= 1e-2 forward, <= 7e-2 for gradients (*); after the ~60-layer model:
<= 3e-2 on z and yh, <= 3e-2 relative on the scalar losses, <= 8e-2 relative L2 on the concatenated gradient.
The fp32 path through the same host wiring is held to 2e-5 / 2e-2 (ReLU-boundary flips, see test_conv_gpu.py).

(*) why gradients are looser than activations: a pre-activation h carries ~3e-3 relative bf16 noise, so the ~0.25 %
of elements with |h| below that noise land on the other side of the ReLU; each flip switches one element of the
upstream gradient fully on or off.  With a random dy every weight / bias gradient is a random-walk sum over rows, so
its relative error equals the element-wise one, sqrt(flipped / active) ~ sqrt(0.0025 / 0.45) ~ 7e-2 at the layer
behind the flips (measured: 4.3e-2 .. 5.3e-2 on K2's gradients, 1e-2 on K1's, 5e-3 on K3's, 2e-3 on dx).  A wiring
error (wrong slice, pitch, dropout key, residual) shows up as O(1).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vqvae_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "speech-masters-thesis_amd")


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def kernel_names():
    from smt_amd import profiler
    return {r["name"] for r in profiler.summary()}


# ------------------------------------------------------------------------------------------------------------
# (a) block level
# ------------------------------------------------------------------------------------------------------------
FAST_FWD = {"conv_k1act:fwd", "conv_ws_pipe:fwd", "conv_ws2:fwd", "conv_k3gate", "conv1x1_c64:fwd"}
FAST_BWD = {"conv_gate_bwd", "gate_mix_bwd", "conv1x1_bwd", "conv_ws:dgrad", "conv_ws2:dgrad", "conv_wgrad_shift", "conv_k1_bwd"}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gated_hifi_w64_fast_kernels_train_mode_vs_oracle(dtype):
    from models.vqvae.resnet import GatedHiFiBlock
    from smt_amd import profiler
    torch.manual_seed(11)
    w, depth, site_base, seed, p_drop = 64, 4, 40, 9, 0.1
    b, t = 3, 24000                                            # 72,000 rows: >= 512 tiles for every branch
    blk = GatedHiFiBlock(w, depth, dilation_growth_rate=3, kernel_size_growth_rate=2, zero_out=False, dropout=p_drop,
                         site_base=site_base).cuda()
    blk.train()
    g = torch.Generator().manual_seed(5)
    x_nct = torch.randn(b, w, t, generator=g).to(dtype).float()         # representable in the compute dtype
    lens = torch.tensor([t, t - 4321, 9001])
    dy_nct = torch.randn(b, w, t, generator=g).to(dtype).float()
    xa = x_nct.permute(0, 2, 1).contiguous().cuda().to(dtype).requires_grad_(True)

    profiler.reset(); profiler.enable(True)
    try:
        y = blk(xa, lens.cuda().to(torch.int32), drop_seed=seed)
        y.backward(dy_nct.permute(0, 2, 1).contiguous().cuda().to(dtype))
        names = kernel_names()
    finally:
        profiler.enable(False); profiler.reset()
    if dtype == torch.bfloat16:
        assert FAST_FWD <= names and FAST_BWD <= names, sorted(names)
        # none of the dilated convs / their gradients fell back to the streaming or generic kernels
        assert not any(n.startswith(("conv_gemm_dma", "conv1x1_dma", "conv_wgrad_dma")) or n == "conv_wgrad" for n in names), sorted(names)
    else:
        assert all(n.split(":")[0] in ("conv_gemm", "conv_wgrad", "conv_wgrad_reduce", "gate_mix_fwd", "gate_mix_bwd") for n in names), sorted(names)

    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    prm = {"b." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ids = {f"b.blocks.{d}.1.drop{s}": site_base + 2 * d + s for d in range(depth) for s in (0, 1)}
    drop = orc.make_counter_dropout(seed, p_drop, ids)
    xr = x_nct.clone().requires_grad_(True)
    mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
    yr = orc.gated_hifi_block(xr, mask, prm, "b", orc.VQVAEConfig(width=w, multipliers=(1, 1, 1)), drop)
    yr.backward(dy_nct)

    fwd_tol, grad_tol = (1e-2, 7e-2) if dtype == torch.bfloat16 else (2e-5, 2e-2)
    errs = {"y": rel_l2(y.permute(0, 2, 1), yr), "dx": rel_l2(xa.grad.permute(0, 2, 1), xr.grad)}
    for name, q in blk.named_parameters():
        errs[name] = rel_l2(q.grad, prm["b." + name].grad)
    print("\n" + "\n".join(f"  {k:32s} {v:.3e}" for k, v in errs.items()))
    assert errs["y"] <= fwd_tol, errs
    bad = {k: v for k, v in errs.items() if k != "y" and v > grad_tol}
    assert not bad, bad
    print(f"\n[w64 {dtype}] rel-L2: y {errs['y']:.2e} dx {errs['dx']:.2e} worst param grad "
          f"{max(v for k, v in errs.items() if k not in ('y', 'dx')):.2e}")


def test_fused_k3_gate_kernel_is_bit_identical_to_the_unfused_path():
    """smt_conv_k3gate_fwd (K3 of the four branches + tanh * softmax gate in one pass) against the four folded K3 launches
    + gate_mix_fwd it replaces: same z, same g -> the block output and every gradient are bit-identical; ragged lens, a
    row count that is not a multiple of the 128-row tile, train mode."""
    from models.vqvae.resnet import GatedHiFiBlock
    from smt_amd import convops, profiler
    torch.manual_seed(3)
    blk = GatedHiFiBlock(64, 4, dilation_growth_rate=3, kernel_size_growth_rate=2, zero_out=False, dropout=0.1,
                         site_base=8).cuda().train()
    g = torch.Generator().manual_seed(9)
    b, t = 3, 5000 + 77
    x = torch.randn(b, t, 64, generator=g).cuda().to(torch.bfloat16)
    dy = torch.randn(b, t, 64, generator=g).cuda().to(torch.bfloat16)
    lens = torch.tensor([t, 4000, 129], dtype=torch.int32).cuda()
    outs = []
    for fused in (True, False):
        convops._K3GATE = fused
        try:
            xa = x.clone().requires_grad_(True)
            blk.zero_grad()
            profiler.reset(); profiler.enable(True)
            y = blk(xa, lens, drop_seed=5)
            y.backward(dy)
            names = kernel_names()
            profiler.enable(False); profiler.reset()
        finally:
            convops._K3GATE = True
        assert ("conv_k3gate" in names) == fused and ("gate_mix_fwd" in names) == (not fused), sorted(names)
        outs.append((y.detach().clone(), xa.grad.clone(), [p.grad.clone() for p in blk.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, c in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, c)


# ------------------------------------------------------------------------------------------------------------
# (b), (c) model level, BASELINE.json configs[1] (codebook 1024, bf16)
# ------------------------------------------------------------------------------------------------------------
def k1024_config(batch):
    from utils import config as C
    return C.merge(C.load(os.path.join(ROOT, "configs/models/vqvae_k1024.yaml")),
                   C.load(os.path.join(ROOT, "configs/datasets/synthetic_ljspeech.yaml")),
                   C.create({"train": {"batch_size": batch, "n_gpus": 1}}))


def build_k1024(batch, seed):
    from models.vqvae.vqvae import VQVAE
    cfg = k1024_config(batch)
    assert cfg.model.compute_dtype == "bf16" and cfg.model.l_bins == 1024
    ocfg = orc.VQVAEConfig.from_dict(k1024_config(batch).model.to_dict())
    # The reference zero-initialises K3 and the gate (blocks start as identities); a freshly initialised model would
    # leave most of the fast path multiplying by zero, and default-scale random values there make a 60-layer chaotic
    # net whose gradients are ill-conditioned.  Quarter-scale values: every kernel contributes, blocks stay near-identity.
    params = orc.init_params(ocfg, seed=seed, zero_out=False)
    for name in params:
        if ".model.5." in name or ".gate." in name:
            params[name] = params[name] * 0.25
    model = VQVAE(cfg).cuda()
    sd = {k: v.clone() for k, v in params.items()}
    sd["bottleneck.level_blocks.0.k"] = torch.zeros(ocfg.l_bins, ocfg.emb_width)
    model.load_state_dict(sd)
    return model, ocfg, params


def test_full_model_bf16_eval_vs_oracle():
    t = 145408
    model, ocfg, params = build_k1024(2, seed=3)
    x = orc.synthetic_clip_batch(2, t, 77)
    lens = torch.tensor([t, 102400])
    # oracle (fp32, CPU): encoder -> codebook from its own rows -> exact codes -> decoder -> losses
    with torch.no_grad():
        x_mask = orc.sequence_mask(lens, t).unsqueeze(1).float()
        z_ref, z_mask = orc.encoder_forward(x, x_mask, params, ocfg)
        rows_ref, mf = orc.vq_preprocess(z_ref, z_mask)
        valid = rows_ref[(mf != 0)[:, 0]]
        k = valid[torch.randperm(valid.shape[0], generator=torch.Generator().manual_seed(0))[:ocfg.l_bins]].contiguous()
    blk = model.bottleneck.level_blocks[0]
    blk.k.copy_(k.cuda()); blk.init = True
    model.eval()
    xc, lc = x.cuda(), lens.cuda()
    with torch.no_grad():
        z, z_lens = model.encoders[0](xc[:, 0], lc.to(torch.int32))
        codes, _ = model.encode_and_quantize(xc, lc)
        loss_dict, metrics = model.supervised_step([None, None, None, None, xc, lc, None])
    assert metrics == {} and z.dtype == torch.bfloat16
    assert z_lens.tolist() == [1136, 800]
    # (1) encoder output
    e_z = rel_l2(z.permute(0, 2, 1), z_ref)
    assert e_z <= 3e-2, e_z
    # (2) code indices: bit-exact vs the exact argmin on the model's OWN z (the index semantics of this build)
    own_rows = z.float().reshape(-1, z.shape[-1]).cpu().numpy()
    exact, _, _ = orc.vq_argmin_exact(own_rows, k.numpy())
    assert np.array_equal(codes.reshape(-1).cpu().numpy(), exact)
    exact_ref, d1, d2 = orc.vq_argmin_exact(rows_ref.numpy(), k.numpy())
    agree = float((exact == exact_ref)[(mf != 0)[:, 0].numpy()].mean())
    assert agree >= 0.85, agree         # bf16 z vs fp32 z pick the same code except on near-ties
    # (3) decoder + losses: the oracle's decoder fed with the product's quantised rows (so that the comparison is
    #     of the decoder arithmetic, not of near-tie code flips), then the end-to-end losses
    with torch.no_grad():
        xq = k[torch.from_numpy(exact)].view(2, -1, ocfg.emb_width).permute(0, 2, 1) * z_mask
        y_ref, _ = orc.decoder_forward(xq, z_mask, params, ocfg)
        rec = orc.multinorm_recon_loss(x, y_ref, x_mask, ocfg)
        stft = orc.multires_stft_loss(x, y_ref, x_mask, ocfg)
        sel = (mf != 0)[:, 0]
        own = torch.from_numpy(own_rows)
        commit = ((k[torch.from_numpy(exact)][sel] - own[sel]) ** 2).sum() / (mf.sum() * ocfg.emb_width)
    e_y = rel_l2(loss_dict["yh"] * x_mask[:, 0].cuda(), y_ref[:, 0] * x_mask[:, 0])
    got = {k_: loss_dict[k_].item() for k_ in ("loss_recon", "loss_stft", "loss_commit", "loss")}
    ref = {"loss_recon": rec.item(), "loss_stft": stft.item(), "loss_commit": commit.item()}
    ref["loss"] = ref["loss_recon"] + ocfg.multispectral * ref["loss_stft"] + ocfg.commit * ref["loss_commit"]
    own = _oracle_losses_at(x, loss_dict["yh"], x_mask, ocfg)
    print(f"\n[k1024 bf16 eval] rel-L2 z {e_z:.2e} yh {e_y:.2e}; code agreement with the fp32 oracle {agree:.4f};\n"
          f"  losses {got}\n  oracle on the oracle's yh {ref}\n  oracle on the product's yh {own}")
    assert e_y <= 3e-2, e_y
    # the loss kernels are fp32: on the product's own yh they must agree with the oracle tightly ...
    assert np.isclose(got["loss_recon"], own["loss_recon"], rtol=1e-4) and np.isclose(got["loss_stft"], own["loss_stft"], rtol=1e-4)
    assert np.isclose(got["loss_commit"], ref["loss_commit"], rtol=1e-4)
    # ... while end to end the LOG-spectral term amplifies bf16's noise floor (see _oracle_losses_at): stated, loose
    assert np.isclose(got["loss_recon"], ref["loss_recon"], rtol=3e-3)
    assert np.isclose(got["loss_stft"], ref["loss_stft"], rtol=0.15)


def test_codes_at_full_c2_size_on_model_rows():
    """N = 36,352 latent rows x K = 1024 (BASELINE.json configs[1], B = 32) produced by the bench model's own encoder:
    every index equals the exact argmin; and the number of rows on which the REFERENCE's fp32 expression
    (sum x^2 - 2 x k^T + sum k^2, bottleneck.py:126-134, torch-CPU sgemm) picks a different code is counted and printed
    (that disagreement is round-off of the reference's expression, never of this build)."""
    from datasets.synthetic import synth_clip
    from smt_amd import vq
    t, b = 145408, 32
    model, ocfg, params = build_k1024(b, seed=5)
    model.eval()
    x = torch.stack([synth_clip(t, 7000 + i) for i in range(b)]).cuda()
    lens = torch.full((b,), t, dtype=torch.int32, device="cuda")
    with torch.no_grad():
        z, z_lens = model.encoders[0](x, lens)
    rows = z.float().reshape(-1, z.shape[-1]).contiguous()
    assert rows.shape == (36352, 128)
    g = torch.Generator().manual_seed(3)
    k = rows[torch.randperm(rows.shape[0], generator=g)[:1024].cuda()].contiguous()     # init_k: codes are data rows
    idx, md, _, sums = vq.vq_forward_raw(rows, k, prep=vq.prepare(k))
    exact, d1, d2 = orc.vq_argmin_exact(rows.cpu().numpy(), k.cpu().numpy())
    assert np.array_equal(idx.cpu().numpy(), exact)
    ref_idx, _, _ = orc.vq_quantize_reference(rows.cpu(), k.cpu())
    differ = int((ref_idx.numpy() != exact).sum())
    rel_gap = (d2 - d1) / np.maximum(d2, 1e-30)
    print(f"\n[C2 size, model rows] exactly re-scored rows {int(sums[3].item())} of 36352; reference fp32 expression differs "
          f"from the exact argmin on {differ} rows (their relative best/runner-up gaps: "
          f"{np.sort(rel_gap[ref_idx.numpy() != exact])[:5]}); rows with relative gap < 1e-5: {int((rel_gap < 1e-5).sum())}")
    # Measured: 74 of 36,352 rows (0.2 %), relative gaps 5e-5 .. 1e-4 -- these rows carry a large common offset, so the
    # reference's sum x^2 - 2 x.k + sum k^2 cancels ~4 digits before the gap shows.  It is round-off of the REFERENCE's
    # expression (this build equals the exact argmin on every row, asserted above); bound it loosely.
    assert differ <= 0.01 * rows.shape[0]


def _oracle_losses_at(x, yh, x_mask, ocfg):
    """The oracle's two waveform losses evaluated at the PRODUCT's yh.  Why: the log-magnitude term of the spectral
    loss, (log|Y| - log max(|Yh|, 1e-5))^2, is ill-conditioned in yh where |Yh| is tiny -- an untrained decoder emits
    almost no energy in the upper bins, and the white ~5e-4 relative rounding noise of the bf16 conv stacks lifts those
    bins by orders of magnitude (measured here: yh agrees to 4e-4 relative L2, loss_recon to 2e-5, loss_stft moves 9 %).
    Parity of the loss KERNELS is therefore asserted on identical inputs, parity of yh separately."""
    with torch.no_grad():
        y = yh.detach().float().cpu().unsqueeze(1)
        return {"loss_recon": orc.multinorm_recon_loss(x, y, x_mask, ocfg).item(),
                "loss_stft": orc.multires_stft_loss(x, y, x_mask, ocfg).item()}


def test_full_size_train_step_bf16_vs_oracle():
    """One TRAIN step of the bench configuration at B = 1 (top-level rows 72,704: weight-stationary kernels at the top
    levels, streaming kernels below), counter dropout ON, injected init / revival rows."""
    from smt_amd import profiler
    t = 145408
    model, ocfg, params = build_k1024(1, seed=4)
    x = orc.synthetic_clip_batch(1, t, 78)
    lens = torch.tensor([t - 6 * 512])
    g = torch.Generator().manual_seed(1)
    k_init = torch.randn(ocfg.l_bins, ocfg.emb_width, generator=g) * 0.05
    k_rand = torch.randn(ocfg.l_bins, ocfg.emb_width, generator=g) * 0.05
    model.train()
    profiler.reset(); profiler.enable(True)
    try:
        loss_dict, metrics = model(x.cuda(), lens.cuda(), k_rand=k_rand.cuda(), k_rand_init=k_init.cuda())
        loss_dict["yh"].retain_grad()
        loss_dict["loss"].backward(retain_graph=True)
        names = kernel_names()
    finally:
        profiler.enable(False); profiler.reset()
    assert (FAST_FWD | FAST_BWD) <= names, sorted(names)
    seed = model._drop_seed

    prm = {n: v.clone().requires_grad_(True) for n, v in params.items()}
    state = orc.CodebookState(k=torch.zeros(ocfg.l_bins, ocfg.emb_width))
    drop = orc.make_counter_dropout(seed, ocfg.dropout, orc.dropout_site_ids(ocfg))
    kr = iter([k_init, k_rand])
    out, m_ref, aux = orc.vqvae_forward(x, lens, prm, ocfg, state, True, drop=drop, k_rand=lambda rows: next(kr))
    x_mask = orc.sequence_mask(lens, t).unsqueeze(1).float()

    # (1) forward
    e_y = rel_l2(loss_dict["yh"], out["yh"])
    own = _oracle_losses_at(x, loss_dict["yh"], x_mask, ocfg)
    got = {k_: loss_dict[k_].item() for k_ in ("loss", "loss_recon", "loss_stft", "loss_commit")}
    print(f"\n[k1024 bf16 train B=1] rel-L2 yh {e_y:.2e}\n  losses {got}\n  oracle e2e "
          f"{ {k_: out[k_].item() for k_ in got} }\n  oracle on the product's yh {own}")
    assert e_y <= 3e-2, e_y
    assert np.isclose(got["loss_recon"], own["loss_recon"], rtol=1e-4) and np.isclose(got["loss_stft"], own["loss_stft"], rtol=1e-4)
    assert np.isclose(got["loss_recon"], out["loss_recon"].item(), rtol=3e-2)
    assert np.isclose(got["loss_commit"], out["loss_commit"].item(), rtol=3e-2)
    assert np.isclose(got["loss_stft"], out["loss_stft"].item(), rtol=0.15)          # log-spectral term, see _oracle_losses_at
    # VQ metrics / codebook: the bf16 z differs from the fp32 z, so code assignments differ on near-ties
    assert np.isclose(float(metrics["fit"]), float(m_ref["fit"]), rtol=3e-2)
    assert np.isclose(float(metrics["entropy"]), float(m_ref["entropy"]), rtol=3e-2, atol=3e-2)
    blk = model.bottleneck.level_blocks[0]
    assert rel_l2(blk.k, state.k) <= 3e-2 and rel_l2(blk.k_elem, state.k_elem) <= 3e-2

    # (2) backward, in two stages so that the ill-conditioned loss does not blur the check of the conv stacks:
    #  (i) dL/dyh of the fp32 loss kernels vs the oracle's autograd AT the product's yh;
    dyh = loss_dict["yh"].grad.detach().cpu()
    yv = loss_dict["yh"].detach().cpu().unsqueeze(1).clone().requires_grad_(True)
    (orc.multinorm_recon_loss(x, yv, x_mask, ocfg) + ocfg.multispectral * orc.multires_stft_loss(x, yv, x_mask, ocfg)).backward()
    e_dyh = rel_l2(dyh, yv.grad[:, 0])
    # fp32 FFT (product) vs fp32 DFT-as-conv (oracle): the log term's 1/|Yh| amplifies their round-off in the decoder's
    # empty upper bins (3.9e-2 here); on the reference's own fixture the same kernels agree to 1e-3 (test_spectral_gpu.py)
    assert e_dyh <= 1e-1, e_dyh
    #  (ii) the network's backward for a well-conditioned upstream gradient: the SAME seeded random dL/dyh (+ the commit
    #       term) pushed through the product graph and through the oracle graph.  (With the spectral loss's own dL/dyh --
    #       high-frequency, 1/|Yh|-amplified -- sums like d out.weight = sum_t dyh[t] h[t, :] cancel almost completely
    #       against the smooth h, and the bf16 storage noise of h alone moves them by 40 %: conditioning, not wiring.)
    named = dict(model.named_parameters())
    e2e = {n: q.grad.detach().cpu().clone() for n, q in named.items()}
    model.zero_grad()
    # + 0.5: keeps sum_t dy[t] (every bias gradient behind the last layer is w * that sum) away from an accidental zero
    dy_inj = torch.randn(1, t, generator=torch.Generator().manual_seed(2)) + 0.5
    torch.autograd.backward([loss_dict["yh"], loss_dict["loss_commit"]],
                            [dy_inj.cuda(), torch.tensor(ocfg.commit, device="cuda")])
    (out["yh"] * dy_inj).sum().add(ocfg.commit * out["loss_commit"]).backward()
    per = sorted(((named[n].grad.cpu() - p.grad).norm().item() / (p.grad.norm().item() + 1e-30), n) for n, p in prm.items())
    num = sum((named[n].grad.cpu() - p.grad).norm().item() ** 2 for n, p in prm.items())
    den = sum(p.grad.norm().item() ** 2 for p in prm.values())
    glob = (num / den) ** 0.5
    print(f"  dL/dyh rel-L2 {e_dyh:.2e}; injected-dy global gradient rel-L2 {glob:.2e}; median tensor "
          f"{per[len(per) // 2][0]:.2e}; worst:\n" + "\n".join(f"    {e:.3e} {n}" for e, n in per[-6:]))
    assert glob <= 8e-2, glob
    assert per[-1][0] <= 0.15, per[-1]
    assert all(torch.isfinite(v).all() for v in e2e.values())


# ------------------------------------------------------------------------------------------------------------
# (d) the reference fixture through the bf16 path
# ------------------------------------------------------------------------------------------------------------
def test_small_golden_eval_step_bf16(golden):
    from test_model_gpu import build, params_from, small_config
    g = golden("vqvae_small")
    model = build(params_from(g, "p."), small_config(l_bins=int(g["k0"].shape[0]), compute_dtype="bf16"))
    blk = model.bottleneck.level_blocks[0]
    blk.k.copy_(torch.from_numpy(g["k0"])); blk.init = True
    model.eval()
    x, lens = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["lens"]).cuda()
    with torch.no_grad():
        z, _ = model.encoders[0](x[:, 0], lens.to(torch.int32))
        codes, _ = model.encode_and_quantize(x, lens)
    exact, _, _ = orc.vq_argmin_exact(z.float().reshape(-1, z.shape[-1]).cpu().numpy(), g["k0"])
    assert np.array_equal(codes.reshape(-1).cpu().numpy(), exact)
    loss_dict, _ = model.supervised_step([None, None, None, None, x, lens, None])
    e_y = rel_l2(loss_dict["yh"], torch.from_numpy(g["eval_yh"]))
    agree = float((codes.cpu().numpy() == g["enc_codes"]).mean())
    got = {kk: loss_dict[kk].item() for kk in ("loss", "loss_recon", "loss_stft")}
    print(f"\n[small fixture, bf16] rel-L2 yh {e_y:.2e}; codes equal to the reference's {agree:.3f}; losses {got} vs "
          f"{ {kk: float(g['eval_' + kk]) for kk in got} }")
    assert agree >= 0.9
    assert e_y <= 3e-2
    assert np.isclose(got["loss_recon"], float(g["eval_loss_recon"]), rtol=3e-3)
    assert np.isclose(got["loss_stft"], float(g["eval_loss_stft"]), rtol=0.15)       # log-spectral term, see _oracle_losses_at
    ocfg = orc.VQVAEConfig(width=16, emb_width=32, l_bins=int(g["k0"].shape[0]), multipliers=(1, 1, 1), linf_topk=128)
    x_mask = orc.sequence_mask(torch.from_numpy(g["lens"]), g["x"].shape[-1]).unsqueeze(1).float()
    own = _oracle_losses_at(torch.from_numpy(g["x"]), loss_dict["yh"], x_mask, ocfg)
    assert np.isclose(got["loss_recon"], own["loss_recon"], rtol=1e-4) and np.isclose(got["loss_stft"], own["loss_stft"], rtol=1e-4)
