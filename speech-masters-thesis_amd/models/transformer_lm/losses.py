"""Alternative training losses of the TransformerLM (reference models/transformer_lm/losses.py): selected by
``config.model.loss_type`` in {"mmi", "focal"}; the default "ce" runs in the native cross-entropy kernel instead
(smt_amd.lm.cross_entropy).  These two are a handful of row-wise torch ops on the [rows, vocab] logits."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class MaximumMutualInformationLoss(nn.Module):
    """H(Z|X) upper bound minus H(Z) (losses.py:8-21).  The reference scores p(z|y) against log_softmax(one_hot(y)), which
    is 1 - c at the target class and -c elsewhere with c = log(e + C - 1); since p sums to one the inner sum collapses to
    p[target] - c, which is what is evaluated here."""

    def __init__(self, num_classes):
        super().__init__()
        self.num_classes = num_classes

    def forward(self, yh, y):
        p = F.softmax(yh, dim=-1)
        marginal = p.mean(0)
        entropy = -(marginal * marginal.log()).sum(-1)
        c = math.log(math.e + self.num_classes - 1)
        cond = c - p.gather(1, y[:, None]).squeeze(1).mean(0)
        return cond - entropy


class FocalLoss(nn.Module):
    """(1 - p_t)^gamma * CE (losses.py:24-103, arXiv 1708.02002) with optional class weights alpha."""

    def __init__(self, gamma=0.0, alpha=None, reduction="mean", ignore_index=-100):
        if reduction not in ("mean", "sum", "none"):
            raise ValueError("Reduction must be one of: 'mean', 'sum', 'none'.")
        super().__init__()
        self.gamma, self.alpha, self.reduction, self.ignore_index = gamma, alpha, reduction, ignore_index

    def __repr__(self):
        return (f"{type(self).__name__}(alpha={self.alpha}, gamma={self.gamma}, ignore_index={self.ignore_index}, "
                f"reduction={self.reduction})")

    def forward(self, x, y):
        if x.ndim > 2:                                   # (N, C, d1..dK) -> (N d1..dK, C)
            x = x.movedim(1, -1).reshape(-1, x.shape[1])
            y = y.reshape(-1)
        keep = y != self.ignore_index
        y = y[keep]
        if y.numel() == 0:
            return 0.
        log_pt = F.log_softmax(x[keep], dim=-1).gather(1, y[:, None]).squeeze(1)
        ce = -log_pt if self.alpha is None else -log_pt * self.alpha.to(log_pt)[y]
        loss = (1 - log_pt.exp()) ** self.gamma * ce
        if self.reduction == "mean":
            return loss.mean()
        return loss.sum() if self.reduction == "sum" else loss
