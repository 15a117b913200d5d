"""Learning-rate schedules selected by ``config.scheduler`` (reference utils/lr_scheduler.py).

The VQ-VAE config has ``scheduler: null`` so only the constant schedule is on the
hot path; the warm-up variants are kept because get_optimizer dispatches to them.
"""
from torch.optim.lr_scheduler import LambdaLR


class DummyLR(LambdaLR):
    """Constant LR (reference lr_scheduler.py:12-16)."""

    def __init__(self, optimizer):
        super().__init__(optimizer, lambda step: 1.0)


class LinearWarmupLR(LambdaLR):
    """Linear ramp to the base LR over ``warmup_steps`` (reference lr_scheduler.py:19-29)."""

    def __init__(self, optimizer, warmup_steps: int):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, lambda step: min((step + 1) / warmup_steps, 1.0))


class NoamLR(LambdaLR):
    """d_model^-0.5 * min(step^-0.5, step * warmup^-1.5) (reference lr_scheduler.py:32-38)."""

    def __init__(self, optimizer, dim_model: int, warmup_steps: int):
        self.dim_model, self.warmup_steps = dim_model, warmup_steps

        def scale(step):
            s = step + 1
            return dim_model ** -0.5 * min(s ** -0.5, s * warmup_steps ** -1.5)

        super().__init__(optimizer, scale)
