"""TextEncoder and FlowSpecDecoder of GlowTTS (reference models/glow_tts/modules.py:9-236) on channels-last activations."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import models.glow_tts.submodules as submodules
from models.glow_tts.submodules import conv
from smt_amd import lm as K


class TextEncoder(nn.Module):
    """modules.py:9-131: embedding, optional prenet, n x [relative-position self-attention, LayerNorm, conv feed-forward,
    LayerNorm], projections to the prior statistics, duration predictor on the detached features."""

    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, filter_channels_dp, n_heads, n_layers, kernel_size,
                 p_dropout, window_size, mean_only=False, prenet=False, gin_channels=0, sites=None):
        super().__init__()
        assert gin_channels == 0, "speaker conditioning is not built (n_speakers = 1 in configs/models/glow_tts.yaml)"
        sites = sites if sites is not None else submodules._Sites()
        self.n_layers, self.hidden_channels, self.prenet, self.mean_only, self.p_dropout = n_layers, hidden_channels, prenet, mean_only, p_dropout
        self.emb = nn.Embedding(n_vocab, hidden_channels)
        nn.init.normal_(self.emb.weight, 0.0, hidden_channels ** -0.5)
        if prenet:
            self.pre = submodules.ConvReluNorm(hidden_channels, hidden_channels, hidden_channels, kernel_size=5, n_layers=3, p_dropout=0.1,
                                               sites=sites, prefix="encoder.pre")
        self.attn_layers, self.norm_layers_1 = nn.ModuleList(), nn.ModuleList()
        self.ffn_layers, self.norm_layers_2 = nn.ModuleList(), nn.ModuleList()
        self.drop_sites = []
        for i in range(n_layers):
            self.attn_layers.append(submodules.AttentionBlock(hidden_channels, hidden_channels, n_heads, window_size=window_size,
                                                              p_dropout=p_dropout, sites=sites, prefix=f"encoder.attn_layers.{i}"))
            self.norm_layers_1.append(submodules.LayerNorm(hidden_channels))
            a = sites.add(f"encoder.drop.attn.{i}")
            self.ffn_layers.append(submodules.FeedForwardNetwork(hidden_channels, hidden_channels, filter_channels, kernel_size,
                                                                 p_dropout=p_dropout, sites=sites, prefix=f"encoder.ffn_layers.{i}"))
            self.norm_layers_2.append(submodules.LayerNorm(hidden_channels))
            self.drop_sites.append((a, sites.add(f"encoder.drop.ffn.{i}")))
        self.proj_m = submodules.ConvParams(hidden_channels, out_channels, 1)
        if not mean_only:
            self.proj_s = submodules.ConvParams(hidden_channels, out_channels, 1)
        self.proj_w = submodules.DurationPredictor(hidden_channels, filter_channels_dp, kernel_size, p_dropout, sites=sites,
                                                   prefix="encoder.proj_w")

    def forward(self, text, text_lengths, seed=0, speaker_embeddings=None):
        """text [B, Tx] int64 -> (x_m [B, Tx, D], x_logs [B, Tx, D] or None when mean_only, logw [B, Tx], lens int32)."""
        assert speaker_embeddings is None
        lens = text_lengths.to(torch.int32)
        x = F.embedding(text, self.emb.weight) * math.sqrt(self.hidden_channels)          # [B, Tx, H], channels-last as it comes
        if self.prenet:
            x = self.pre(x, lens, seed)
        for i in range(self.n_layers):
            sa, sf = self.drop_sites[i]
            y = self.attn_layers[i](x, lens, seed)
            x = self.norm_layers_1[i](x, y, K.Drop(self.p_dropout, self.training, seed, sa))
            y = self.ffn_layers[i](x, lens, seed)
            x = self.norm_layers_2[i](x, y, K.Drop(self.p_dropout, self.training, seed, sf))
        x_m = conv(x, self.proj_m.weight, self.proj_m.bias, lens=lens)
        x_logs = None if self.mean_only else conv(x, self.proj_s.weight, self.proj_s.bias, lens=lens)
        logw = self.proj_w(x.detach(), lens, seed)
        return x_m, x_logs, logw, lens


class FlowSpecDecoder(nn.Module):
    """modules.py:134-236: n_blocks x [ActNorm, InvConvNear, CouplingBlock] on the n_sqz-squeezed spectrogram."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_blocks, n_layers, p_dropout=0.0, n_split=4, n_sqz=2,
                 sigmoid_scale=False, gin_channels=0, sites=None):
        super().__init__()
        sites = sites if sites is not None else submodules._Sites()
        self.n_sqz = n_sqz
        self.flows = nn.ModuleList()
        for blk in range(n_blocks):
            self.flows.append(submodules.ActNorm(channels=in_channels * n_sqz))
            self.flows.append(submodules.InvConvNear(channels=in_channels * n_sqz, n_split=n_split))
            self.flows.append(submodules.CouplingBlock(in_channels * n_sqz, hidden_channels, kernel_size=kernel_size,
                                                       dilation_rate=dilation_rate, n_layers=n_layers, p_dropout=p_dropout,
                                                       sigmoid_scale=sigmoid_scale, gin_channels=gin_channels, sites=sites,
                                                       prefix=f"decoder.flows.{3 * blk + 2}"))

    def forward(self, spect, lens, speaker_embeddings=None, reverse=False, seed=0):
        """spect [B, T, n_mels] channels-last, lens [B] (a multiple of n_sqz each) -> (z [B, T, n_mels], logdet [B] or None)."""
        x, xl = self.squeeze(spect, lens, self.n_sqz) if self.n_sqz > 1 else (spect, lens)
        logdet_tot = None if reverse else 0
        for f in (reversed(self.flows) if reverse else self.flows):
            x, logdet = f(x, xl, reverse=reverse, seed=seed)
            if not reverse:
                logdet_tot = logdet_tot + logdet
        if self.n_sqz > 1:
            x = self.unsqueeze(x, self.n_sqz)
        return x, logdet_tot

    @staticmethod
    def squeeze(x, lens, n_sqz=2):
        """[B, T, C] -> [B, T / n, n C] with channel order (frame offset k, mel c) -> k C + c: in channels-last rows this is a
        VIEW (n consecutive frames side by side); the mask keeps every n-th step (modules.py:205-218).  Rows beyond the length
        are zeroed by the first flow (ActNorm masks its output), as `x_sqz * x_mask` does in the reference."""
        b, t, c = x.shape
        t = (t // n_sqz) * n_sqz
        return x[:, :t].reshape(b, t // n_sqz, n_sqz * c), lens // n_sqz

    @staticmethod
    def unsqueeze(x, n_sqz=2):
        b, t, c = x.shape
        return x.reshape(b, t * n_sqz, c // n_sqz)

    def store_inverse(self):
        for f in self.flows:
            f.store_inverse()
