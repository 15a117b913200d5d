"""TEST INFRASTRUCTURE -- CPU restatement of GlowTTS's monotonic alignment search (reference
models/glow_tts/submodules.py:28-67, `maximum_path`), numpy, line for line except `np.bool` -> `bool` (the alias was removed
from numpy).  Pinned by tests/golden/mas.npz, captured from the reference's own function (tests/golden/make_golden.py: mas).
Only tests/ may import this module."""
import numpy as np


def maximum_path(value: np.ndarray, mask: np.ndarray, max_neg_val=None) -> np.ndarray:
    if max_neg_val is None:
        max_neg_val = -np.inf
    value = (value * mask).astype(np.float32)
    mask = mask.astype(bool)
    b, t_x, t_y = value.shape
    direction = np.zeros(value.shape, dtype=np.int64)
    v = np.zeros((b, t_x), dtype=np.float32)
    x_range = np.arange(t_x, dtype=np.float32).reshape(1, -1)
    for j in range(t_y):
        v0 = np.pad(v, [[0, 0], [1, 0]], mode="constant", constant_values=max_neg_val)[:, :-1]
        v1 = v
        max_mask = v1 >= v0
        v_max = np.where(max_mask, v1, v0)
        direction[:, :, j] = max_mask
        index_mask = x_range <= j
        v = np.where(index_mask, v_max + value[:, :, j], max_neg_val).astype(np.float32)
    direction = np.where(mask, direction, 1)
    path = np.zeros(value.shape, dtype=np.float32)
    index = mask[:, :, 0].sum(1).astype(np.int64) - 1
    index_range = np.arange(b)
    for j in reversed(range(t_y)):
        path[index_range, index, j] = 1
        index = index + direction[index_range, index, j] - 1
    return path * mask.astype(np.float32)
