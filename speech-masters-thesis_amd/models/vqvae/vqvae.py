"""Waveform VQ-VAE (reference models/vqvae/vqvae.py:11-132): encoder -> vector-quantised
bottleneck with EMA codebook -> decoder, trained with a multi-norm reconstruction loss, a
multi-resolution spectral loss and the commitment loss.

Construction follows the reference, including its "keep only the last level" hack that
rewrites ``config.model.levels`` / ``multipliers`` in place (vqvae.py:65-70) -- downstream
tools read the mutated config.  Internally activations are channels-last [B, T, C].
"""
import torch
import torch.nn as nn

from smt_amd import convops
from models.base import WaveformReconstructionModel
from models.vqvae.bottleneck import Bottleneck
from models.vqvae.encdec import Decoder, Encoder
from models.vqvae.losses import MultiNormReconstructionLoss, MultiResolutionSpectralLoss


class VQVAE(WaveformReconstructionModel):

    LEVEL = -1

    def __init__(self, config):
        super().__init__()
        m = config.model
        multipliers = list(m.multipliers) if m.get("multipliers") is not None else [1] * m.levels
        assert len(multipliers) == m.levels, "Invalid number of multipliers"
        if not m.get("use_bottleneck", True):
            raise ValueError("use_bottleneck=false has no native path; the hot path is the quantised model")

        # The reference builds every level and then keeps only the last one; build that one directly.
        level = m.levels - 1 if VQVAE.LEVEL == -1 else VQVAE.LEVEL
        block_kwargs = dict(
            width=m.width * multipliers[level], depth=m.depth * multipliers[level], m_conv=m.get("m_conv", 1.0),
            dilation_growth_rate=m.dilation_growth_rate, dilation_cycle=m.get("dilation_cycle"),
            kernel_size_growth_rate=m.kernel_size_growth_rate, kernel_size_cycle=m.get("kernel_size_cycle"),
            zero_out=m.zero_out, dropout=m.get("dropout", 0.1))
        encoder = Encoder(1, m.emb_width, level + 1, m.downs_t[:level + 1], m.strides_t[:level + 1],
                          m.block_type, site_base=0, **block_kwargs)
        decoder = Decoder(1, m.emb_width, level + 1, m.downs_t[:level + 1], m.strides_t[:level + 1],
                          m.block_type, site_base=encoder.n_sites,
                          reverse_decoder_dilation=m.get("reverse_decoder_dilation", False), **block_kwargs)
        self.encoders = nn.ModuleList([encoder])
        self.decoders = nn.ModuleList([decoder])
        config.model.levels = 1
        config.model.multipliers = [multipliers[level]]
        self.levels = 1

        self.bottleneck = Bottleneck(m.l_bins, m.emb_width, m.mu, 1, m.revival_threshold)
        loss = m.loss
        self.multi_stft_loss = MultiResolutionSpectralLoss(n_ffts=loss.n_ffts, hop_lengths=loss.hop_lengths,
                                                           win_lengths=loss.win_lengths, window=loss.window,
                                                           log=loss.log)
        self.multi_recon_loss = MultiNormReconstructionLoss(l1=loss.l1, l2=loss.l2, linf=loss.linf,
                                                            linf_topk=loss.linf_topk)
        self.commit, self.multispectral = loss.commit, loss.multispectral
        self.compute_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[m.get("compute_dtype", "fp32")]
        for stage in encoder.level_blocks:
            stage.act_dtype = self.compute_dtype
        self._drop_seed = 0
        self._seed_dev = self._keys_dev = None     # device-resident dropout keys (enable_device_keys)
        self._n_sites = encoder.n_sites + decoder.n_sites

    def enable_device_keys(self, flag=True):
        """Keep the dropout step counter and the per-site keys in device memory, refreshed at the top of every training
        forward (smt_lm_make_keys), instead of passing keys by value: a captured hipGraph of the step then draws fresh
        masks on every replay (smt_amd/graph.py).  Same masks either way: host and device counters advance together."""
        if not flag:
            self._seed_dev = self._keys_dev = None
            return
        dev = next(self.parameters()).device
        self._seed_dev = torch.tensor([self._drop_seed & 0x7FFFFFFF], dtype=torch.int32, device=dev)
        self._keys_dev = torch.zeros(max(8, self._n_sites + 8), dtype=torch.int32, device=dev)
        convops.make_device_keys(self._seed_dev, self._keys_dev)

    # Checkpoints written by the reference carry the six DFT-basis buffers of the loss
    # (multi_stft_loss.stfts.N.{forward,inverse}_basis); they are derived data here.
    def load_state_dict(self, state_dict, strict=True, **kw):
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith("_basis")}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    @convops.forward_scope
    def forward(self, x, x_lengths, speaker=None, **vq_kwargs):
        """x [B, 1, T] float in [-1, 1]; x_lengths [B] -> (loss_dict, vq metrics)."""
        b, c, t = x.shape
        assert c == 1
        lens = x_lengths.to(torch.int32)
        self._drop_seed += 1
        if self._keys_dev is not None:                 # the same counter on the device, and this step's keys derived from it
            self._seed_dev.add_(1)
            convops.make_device_keys(self._seed_dev, self._keys_dev)
        target = x.reshape(b, t)
        previous, convops.DEVICE_KEYS = convops.DEVICE_KEYS, (self._keys_dev if self.training else None)
        try:
            z, z_lens = self.encoders[0](target, lens, self._drop_seed)
            _, xqs, commits, vq_metrics = self.bottleneck([z.float()], [z_lens], **vq_kwargs)
            y, _ = self.decoders[0](xqs[0].to(self.compute_dtype), z_lens, self._drop_seed)
        finally:
            convops.DEVICE_KEYS = previous
        assert y.shape == (b, t), f"Expected shape {(b, t)}, got {tuple(y.shape)}."
        loss_recon = self.multi_recon_loss(target, y, lens)
        loss_stft = self.multi_stft_loss(target, y, lens)
        loss_commit = sum(commits)
        loss = loss_recon + self.multispectral * loss_stft + self.commit * loss_commit
        out = {"loss": loss, "loss_recon": loss_recon, "loss_stft": loss_stft, "loss_commit": loss_commit, "yh": y}
        return out, (vq_metrics[-1] if self.training else {})

    @torch.no_grad()
    def encode_and_quantize(self, x, x_lengths):
        """Encode-only pass of scripts/generate_vq_dataset.py:61-70."""
        b, _, t = x.shape
        z, z_lens = self.encoders[0](x.reshape(b, t), x_lengths.to(torch.int32))
        return self.bottleneck.level_blocks[0].encode(z.float(), z_lens), z_lens

    @torch.no_grad()
    def dequantize_and_decode(self, q, q_lengths):
        """codes [B, T'] + lengths -> masked reconstruction [B, 1, T] (scripts/generate_vq_dataset.py:72-80)."""
        z_lens = q_lengths.to(torch.int32)
        xq = self.bottleneck.level_blocks[0].decode(q)
        keep = (torch.arange(q.shape[1], device=q.device)[None, :] < z_lens[:, None]).unsqueeze(-1)
        y, y_lens = self.decoders[0]((xq * keep).to(self.compute_dtype), z_lens)
        keep_t = torch.arange(y.shape[1], device=y.device)[None, :] < y_lens[:, None]
        return (y * keep_t).unsqueeze(1)
