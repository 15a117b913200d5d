"""Down- / up-sampling stages of the encoder and decoder (reference models/vqvae/conv.py).

Each stage owns a ``blocks`` ModuleList with the reference's element order so the
state-dict keys ``...level_blocks.N.blocks.I.*`` line up:
  encoder:  [strided conv k=2s, GatedHiFi] * down_t, conv k3        (conv.py:38-84)
  decoder:  conv k3, [GatedHiFi, transposed conv k=2s] * down_t     (conv.py:87-143)
Row masks are prefix masks, carried as per-item lengths: a stride-s conv maps
L -> ceil(L/s) (``mask[:, :, ::s]``, conv.py:9) and a transposed conv L -> L*s
(``repeat_interleave``, conv.py:17).
"""
import torch
import torch.nn as nn

from smt_amd import convops
from models.vqvae.resnet import ConvParams, GatedHiFiBlock


def get_block(block_type):
    if block_type == "gated_hifi":
        return GatedHiFiBlock
    raise ValueError(f"block_type={block_type!r}: only 'gated_hifi' has a native implementation "
                     "(the one configs/models/vqvae.yaml uses)")


class EncoderConvBlock(nn.Module):
    def __init__(self, input_emb_width, output_emb_width, down_t, stride_t, block_type, width, depth, m_conv=1.0,
                 site_base=0, **block_kwargs):
        super().__init__()
        self.stride_t, self.down_t = stride_t, down_t
        Block = get_block(block_type)
        blocks = []
        for i in range(down_t):
            blocks.append(ConvParams(input_emb_width if i == 0 else width, width, 2 * stride_t))
            blocks.append(Block(width, depth, site_base=site_base + i * 2 * depth, **block_kwargs))
        if down_t > 0:
            blocks.append(ConvParams(width, output_emb_width, 3))
        self.blocks = nn.ModuleList(blocks)
        self.n_sites = down_t * 2 * depth
        self.act_dtype = torch.float32

    @convops.forward_scope
    def forward(self, x, lens, drop_seed=0):
        s = self.stride_t
        for i in range(self.down_t):
            conv, block = self.blocks[2 * i], self.blocks[2 * i + 1]
            if conv.c_in == 1:   # raw waveform [B, T] fp32 -> first feature map
                x = convops.conv_in(x, conv.weight, conv.bias, stride=s, padding=s // 2, lens=lens,
                                    out_dtype=self.act_dtype)
            else:
                x = convops.conv1d(x, conv.weight, conv.bias, stride=s, padding=s // 2, lens=lens)
            lens = (lens + s - 1) // s
            x = block(x, lens, drop_seed)
        if self.down_t > 0:
            conv = self.blocks[-1]
            x = convops.conv1d(x, conv.weight, conv.bias, padding=1, lens=lens)
        return x, lens


class DecoderConvBlock(nn.Module):
    def __init__(self, input_emb_width, output_emb_width, down_t, stride_t, block_type, width, depth, m_conv=1.0,
                 site_base=0, reverse_decoder_dilation=False, **block_kwargs):
        super().__init__()
        self.stride_t, self.down_t = stride_t, down_t
        Block = get_block(block_type)
        blocks = []
        if down_t > 0:
            blocks.append(ConvParams(output_emb_width, width, 3))
            for i in range(down_t):
                blocks.append(Block(width, depth, site_base=site_base + i * 2 * depth, **block_kwargs))
                blocks.append(ConvParams(width, input_emb_width if i == down_t - 1 else width, 2 * stride_t,
                                         transposed=True))
        self.blocks = nn.ModuleList(blocks)
        self.n_sites = down_t * 2 * depth

    @convops.forward_scope
    def forward(self, x, lens, drop_seed=0):
        s = self.stride_t
        if self.down_t > 0:
            conv = self.blocks[0]
            x = convops.conv1d(x, conv.weight, conv.bias, padding=1, lens=lens)
        for i in range(self.down_t):
            block, up = self.blocks[1 + 2 * i], self.blocks[2 + 2 * i]
            x = block(x, lens, drop_seed)
            x = convops.conv_transpose1d(x, up.weight, up.bias, stride=s, padding=s // 2, lens=lens)
            lens = lens * s
        return x, lens
