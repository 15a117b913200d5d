"""Multi-level encoder / decoder (reference models/vqvae/encdec.py:6-83) on channels-last
activations.  The decoder's final 1x1 projection acts on masked rows (encdec.py:82)."""
import torch.nn as nn

from smt_amd import convops
from models.vqvae.conv import DecoderConvBlock, EncoderConvBlock
from models.vqvae.resnet import ConvParams


class Encoder(nn.Module):
    def __init__(self, input_emb_width, output_emb_width, levels, downs_t, strides_t, block_type, site_base=0,
                 **block_kwargs):
        super().__init__()
        self.input_emb_width, self.output_emb_width, self.levels = input_emb_width, output_emb_width, levels
        self.downs_t, self.strides_t = list(downs_t), list(strides_t)
        block_kwargs.pop("reverse_decoder_dilation", None)
        stages, base = [], site_base
        for level, (down_t, stride_t) in enumerate(zip(self.downs_t, self.strides_t)):
            stage = EncoderConvBlock(input_emb_width if level == 0 else output_emb_width, output_emb_width, down_t,
                                     stride_t, block_type, site_base=base, **block_kwargs)
            base += stage.n_sites
            stages.append(stage)
        self.level_blocks = nn.ModuleList(stages)
        self.n_sites = base - site_base

    @convops.forward_scope
    def forward(self, x, lens, drop_seed=0):
        """x: [B, T] fp32 waveform when input_emb_width == 1, else [B, T, C]."""
        b, t = x.shape[0], x.shape[1]
        assert (x.dim() == 2) == (self.input_emb_width == 1)
        for stage, down_t, stride_t in zip(self.level_blocks, self.downs_t, self.strides_t):
            x, lens = stage(x, lens, drop_seed)
            t = t // (stride_t ** down_t)
            assert x.shape == (b, t, self.output_emb_width), f"expected {(b, t, self.output_emb_width)}, got {tuple(x.shape)}"
        return x, lens


class Decoder(nn.Module):
    def __init__(self, input_emb_width, output_emb_width, levels, downs_t, strides_t, block_type="gated_hifi",
                 site_base=0, **block_kwargs):
        super().__init__()
        self.input_emb_width, self.output_emb_width, self.levels = input_emb_width, output_emb_width, levels
        self.downs_t, self.strides_t = list(downs_t), list(strides_t)
        # sites are numbered in execution order: deepest level first
        stages, base = [None] * levels, site_base
        for level in reversed(range(levels)):
            stage = DecoderConvBlock(output_emb_width, output_emb_width, self.downs_t[level], self.strides_t[level],
                                     block_type, site_base=base, **block_kwargs)
            base += stage.n_sites
            stages[level] = stage
        self.level_blocks = nn.ModuleList(stages)
        self.out = ConvParams(output_emb_width, input_emb_width, 1)
        self.n_sites = base - site_base

    @convops.forward_scope
    def forward(self, x, lens, drop_seed=0):
        b, t, c = x.shape
        assert c == self.output_emb_width
        for level in reversed(range(self.levels)):
            x, lens = self.level_blocks[level](x, lens, drop_seed)
            t = t * (self.strides_t[level] ** self.downs_t[level])
            assert x.shape == (b, t, self.output_emb_width), f"expected {(b, t, self.output_emb_width)}, got {tuple(x.shape)}"
        assert self.input_emb_width == 1
        y = convops.conv_out(x.contiguous(), self.out.weight, self.out.bias, lens=lens)   # fp32 [B, T]
        return y, lens
