import torch


def safe_log(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Clamped logarithm used by the spectral losses and VQ entropy (reference utils/torch_utils.py:4-5)."""
    return x.clamp(min=eps).log()
