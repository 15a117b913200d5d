"""GlowTTS (reference models/glow_tts/glow_tts.py:12-168): text encoder -> prior statistics, flow decoder -> latent, monotonic
alignment search between them ON THE DEVICE (the reference round-trips through numpy every step, glow_tts.py:87-97), MLE and
duration losses.  Single speaker.  Activations are channels-last; the public tensors keep the reference's layouts
(spectrograms [B, n_mels, T])."""
import math

import torch
import torch.nn as nn

import models.glow_tts.submodules as submodules
from models.base import TokenToSpectrogramModel
from models.glow_tts.modules import FlowSpecDecoder, TextEncoder
from smt_amd import glow


class GlowTTS(TokenToSpectrogramModel):

    def __init__(self, config):
        super().__init__()
        m, ds = config.model, config.dataset
        if m.n_speakers > 1:
            raise ValueError("n_speakers > 1 (speaker embeddings) has no native path; configs/models/glow_tts.yaml is single-speaker")
        e, d = m.encoder, m.decoder
        self.sites = submodules._Sites()
        self.encoder = TextEncoder(n_vocab=e.n_vocab + int(bool(ds.get("intersperse_blanks", False))), out_channels=ds.n_mels,
                                   hidden_channels=e.hidden_channels, filter_channels=e.filter_channels,
                                   filter_channels_dp=e.filter_channels,          # as the reference passes it (glow_tts.py:27)
                                   n_heads=e.n_heads, n_layers=e.n_layers, kernel_size=e.kernel_size, p_dropout=e.p_dropout,
                                   window_size=e.window_size, mean_only=e.mean_only, prenet=e.prenet, gin_channels=m.gin_channels,
                                   sites=self.sites)
        self.decoder = FlowSpecDecoder(in_channels=ds.n_mels, hidden_channels=d.hidden_channels, kernel_size=d.kernel_size,
                                       dilation_rate=d.dilation_rate, n_blocks=d.n_blocks, n_layers=d.n_layers, p_dropout=d.p_dropout,
                                       n_split=d.n_split, n_sqz=d.n_sqz, sigmoid_scale=d.sigmoid_scale, gin_channels=m.gin_channels,
                                       sites=self.sites)
        self._drop_seed = 0

    def dropout_sites(self):
        """Site name -> id of every dropout in forward order (the oracle replays the masks by name)."""
        return {n: i for i, n in enumerate(self.sites.names)}

    @torch.no_grad()
    def ddi(self, batch):
        """Data-dependent initialisation of the ActNorm layers (glow_tts.py:49-56)."""
        self.train()
        for f in self.decoder.flows:
            if getattr(f, "set_ddi", False):
                f.set_ddi(True)
        _ = self.supervised_step(batch)

    def forward(self, x, x_lengths, y, y_lengths, speaker=None, noise=None):
        """x [B, Tx] tokens, y [B, n_mels, Ty] log-mels -> ({loss_mle, loss_length, loss, yh}, {})."""
        assert speaker is None
        self._drop_seed += 1
        seed, n_sqz = self._drop_seed, self.decoder.n_sqz
        if x_lengths is None:
            x_lengths = torch.full((x.shape[0],), x.shape[1], device=x.device)
        x_m, x_logs, logw_enc, x_lens = self.encoder(x, x_lengths, seed)
        y_max = (y.size(2) // n_sqz) * n_sqz
        if y_lengths is None:
            y_lengths = torch.full((y.shape[0],), y_max, device=y.device)
        y_lens = ((y_lengths // n_sqz) * n_sqz).to(torch.int32)
        spect = y[:, :, :y_max].transpose(1, 2).contiguous().float()                      # [B, Ty, n_mels]
        z_dec, logdet = self.decoder(spect, y_lens, reverse=False, seed=seed)

        # monotonic alignment search on the device: prior log-likelihood -> smt_maximum_path -> frame -> token index
        with torch.no_grad():
            logp = glow.prior_logp(x_m, x_logs, z_dec)
            tx, ty = logp.shape[1], logp.shape[2]
            attn_mask = (submodules.sequence_mask(x_lens, tx).unsqueeze(-1) & submodules.sequence_mask(y_lens, ty).unsqueeze(1)).float()
            attn = submodules.maximum_path(logp, attn_mask)
            idx, durations = glow.align_index(attn)
        z_m = glow.align_gather(x_m, idx)
        z_logs = None if x_logs is None else glow.align_gather(x_logs, idx)

        yh = None
        if not self.training:
            with torch.no_grad():
                w = durations * submodules.sequence_mask(x_lens, tx).float()
                z_lens = ((torch.clamp_min(w.sum(1), 1).long() // n_sqz) * n_sqz).to(torch.int32)
                eps = torch.randn_like(z_m) if noise is None else noise.transpose(1, 2).contiguous()
                t_out = int(z_lens.max())              # sequence_mask(z_lengths, None): the mask is as long as the longest item
                z_mask = submodules.sequence_mask(z_lens, t_out).unsqueeze(-1).float()
                z_enc = ((z_m + (torch.exp(z_logs) if z_logs is not None else 1.0) * eps)[:, :t_out] * z_mask).contiguous()
                yh_rows, _ = self.decoder(z_enc, z_lens, reverse=True)
                yh = yh_rows.transpose(1, 2)
        denom = (y_lens.sum() * z_dec.shape[2]).float()
        l_mle = glow.mle_loss(z_dec, z_m, z_logs, torch.sum(logdet), denom)
        l_length = glow.length_loss(logw_enc, durations, x_lens, x_lengths.sum().float())
        return {"loss_mle": l_mle, "loss_length": l_length, "loss": l_mle + l_length, "yh": yh}, {}
