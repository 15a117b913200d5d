"""Parameter EMA (reference models/ema.py:11-66): fp32 shadow of the trainable
parameters, ``state = mu*state + (1-mu)*p`` after every optimiser step, swapped in
around validation.  The update is one fused multi-tensor lerp over all parameters."""
import logging

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


class DummyEMA(nn.Module):
    """No-op stand-in so the train loop is branch-free (reference ema.py:11-21)."""

    def forward(self, *args, **kwargs):
        raise NotImplementedError("EMA modules have no forward pass")

    def step(self):
        return None

    def swap(self):
        return None


class EMA(nn.Module):
    def __init__(self, model, mu=0.99):
        super().__init__()
        self.mu = mu
        self._names, self._params, self._shadow = [], [], []
        for name, p in model.named_parameters():
            if p.requires_grad:
                self._names.append(name)
                self._params.append(p)
                self._shadow.append(p.detach().float().clone())

    def forward(self, *args, **kwargs):
        raise NotImplementedError("EMA modules have no forward pass")

    def state_dict(self, *args, **kwargs):
        return dict(zip(self._names, self._shadow))

    def load_state_dict(self, state_dict, *args, **kwargs):
        missing = sorted(set(self._names) - set(state_dict))
        if missing:
            logger.warning("%d parameters absent from the EMA state: %s", len(missing), ", ".join(missing))
        index = {n: i for i, n in enumerate(self._names)}
        for name, value in state_dict.items():
            assert name in index, f"EMA state has {name}, the model does not"
            i = index[name]
            assert self._shadow[i].shape == value.shape, f"{name}: {tuple(value.shape)} vs {tuple(self._shadow[i].shape)}"
            self._shadow[i] = value.to(self._shadow[i].device, torch.float32).clone()

    @torch.no_grad()
    def step(self):
        # shadow += (1 - mu) * (p - shadow)  ==  mu*shadow + (1-mu)*p
        torch._foreach_lerp_(self._shadow, [p.detach().float() for p in self._params], 1.0 - self.mu)

    @torch.no_grad()
    def swap(self):
        for p, s in zip(self._params, self._shadow):
            tmp = p.detach().float().clone()
            p.copy_(s.to(p.dtype))
            s.copy_(tmp)
        from smt_amd import convops
        convops.invalidate_packed_weights()       # belt and braces: copy_ bumps the version, foreign writers might not
