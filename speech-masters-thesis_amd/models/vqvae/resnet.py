"""Gated multi-receptive-field residual block of the VQ-VAE encoder / decoder.

Mirrors the parameter tree of the reference ``GatedHiFiBlock`` (models/vqvae/
resnet.py:184-241; per branch ``Sequential(Conv1d 1x1, ResLayer)``, resnet.py:16-36)
so checkpoints interchange, but activations are channels-last and the arithmetic goes
through ``smt_amd.convops``.  Only ``block_type: gated_hifi`` is implemented natively --
it is the only one configs/models/vqvae.yaml exercises.
"""
import math

import torch
import torch.nn as nn

from smt_amd import convops


class ConvParams(nn.Module):
    """weight / bias of one (transposed) 1-D convolution in torch's layouts and default init."""

    def __init__(self, c_in, c_out, kernel, transposed=False, zero=False):
        super().__init__()
        shape = (c_in, c_out, kernel) if transposed else (c_out, c_in, kernel)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(c_out))
        self.c_in, self.c_out, self.kernel, self.transposed = c_in, c_out, kernel, transposed
        self.reset_parameters(zero)

    @torch.no_grad()
    def reset_parameters(self, zero=False):
        if zero:
            self.weight.zero_()
            self.bias.zero_()
            return
        bound = 1.0 / math.sqrt(self.weight.shape[1] * self.weight.shape[2])  # torch's fan_in rule
        self.weight.uniform_(-bound, bound)
        self.bias.uniform_(-bound, bound)


def _numbered(**children):
    holder = nn.Module()
    for name, child in children.items():
        holder.add_module(name.lstrip("_"), child)
    return holder


def mod_cycle(depth, cycle):
    return depth if cycle is None else depth % cycle


class GatedHiFiBlock(nn.Module):
    def __init__(self, n_in, n_depth, dilation_growth_rate=1, dilation_cycle=None, kernel_size_growth_rate=2,
                 kernel_size_cycle=None, zero_out=True, res_scale=False, dropout=0.1, site_base=0, **unused):
        super().__init__()
        self.n_in, self.n_depth, self.dropout = n_in, n_depth, dropout
        self.res_scale = 1.0 if not res_scale else 1.0 / math.sqrt(n_depth)
        self.site_base = site_base
        self.geometry = []
        branches = []
        for d in range(n_depth):
            dil = dilation_growth_rate ** mod_cycle(d, dilation_cycle)
            k = 3 + kernel_size_growth_rate * mod_cycle(d, kernel_size_cycle)
            self.geometry.append((k, dil, ((k - 1) * dil) // 2))
            res = _numbered(model=_numbered(_2=ConvParams(2 * n_in, 2 * n_in, k),
                                            _5=ConvParams(2 * n_in, 2 * n_in, 1, zero=zero_out)))
            branches.append(_numbered(_0=ConvParams(n_in, 2 * n_in, 1), _1=res))
        self.blocks = nn.ModuleList(branches)
        self.gate = ConvParams(n_in, n_in, 1, zero=zero_out)

    @convops.forward_scope
    def forward(self, x, lens, drop_seed=0):
        """x [B, T, n_in]; lens [B] valid lengths (the block's row mask)."""
        if self.res_scale != 1.0:
            raise NotImplementedError("res_scale=True is not used by the reference configs (conv.py:55)")
        params = []
        for branch in self.blocks:
            expand, res = getattr(branch, "0"), getattr(branch, "1").model
            conv_k, conv_1 = getattr(res, "2"), getattr(res, "5")
            params += [expand.weight, expand.bias, conv_k.weight, conv_k.bias, conv_1.weight, conv_1.bias]
        params += [self.gate.weight, self.gate.bias]
        return convops.gated_hifi_block(x, lens, self.geometry, params, p_drop=self.dropout, training=self.training,
                                        seed=drop_seed, site_base=self.site_base)
