"""Monotonic alignment search on the device (smt_maximum_path) against the reference's golden path and the
numpy oracle: the 0/1 alignment must be identical."""
import numpy as np
import pytest
import torch

from oracle import mas_oracle

pytestmark = pytest.mark.gpu


def test_maximum_path_matches_reference_golden(golden):
    from models.glow_tts.submodules import maximum_path
    g = golden("mas")
    value, mask = torch.from_numpy(g["value"]).cuda(), torch.from_numpy(g["mask"]).cuda()
    path = maximum_path(value, mask)
    assert path.dtype == value.dtype and path.device == value.device
    assert np.array_equal(path.cpu().numpy(), g["path"])


@pytest.mark.parametrize("b,t_x,t_y,seed", [(32, 200, 870, 0), (3, 300, 301, 1), (2, 1, 7, 2), (5, 64, 64, 3), (2, 513, 600, 4)])
def test_maximum_path_matches_numpy_oracle(b, t_x, t_y, seed):
    """LJSpeech-like sizes (<= 200 tokens x 870 mel frames per item), more than 256 rows (several rows per thread), one
    token, square."""
    from models.glow_tts.submodules import maximum_path
    g = torch.Generator().manual_seed(seed)
    value = torch.randn(b, t_x, t_y, generator=g) * 2.0
    x_len = torch.randint(1, t_x + 1, (b,), generator=g); x_len[0] = t_x
    y_len = torch.maximum(torch.randint(1, t_y + 1, (b,), generator=g), x_len.clamp(max=t_y)); y_len[0] = t_y
    mask = ((torch.arange(t_x)[None, :, None] < x_len[:, None, None]) &
            (torch.arange(t_y)[None, None, :] < y_len[:, None, None])).float()
    ref = mas_oracle.maximum_path(value.numpy(), mask.numpy())
    got = maximum_path(value.cuda(), mask.cuda()).cpu().numpy()
    assert np.array_equal(got, ref)
    assert got[0].sum() == t_y                                     # every frame of the full item is aligned


def test_backtrack_stays_in_bounds_where_the_reference_raises():
    """with NaN log-likelihoods, or an empty first mask column and t_y > t_x, the backtrack index keeps falling;
    numpy wraps a negative index once and raises IndexError beyond -t_x (models/glow_tts/submodules.py:62-66).  The kernel
    must stop there instead of reading its LDS bitmaps and writing `path` out of bounds: the output stays a 0/1 matrix
    inside the mask, the neighbouring batch items are untouched, nothing faults."""
    from models.glow_tts.submodules import maximum_path
    b, t_x, t_y = 3, 5, 40
    g = torch.Generator().manual_seed(0)
    value = torch.randn(b, t_x, t_y, generator=g)
    mask = torch.ones(b, t_x, t_y)
    value[1] = float("nan")                       # v1 >= vprev is false everywhere: the index falls by one per column
    mask[2, :, 0] = 0.0                           # index starts at -1 and t_y > t_x
    with pytest.raises(IndexError):
        mas_oracle.maximum_path(value[1:2].numpy(), mask[1:2].numpy())
    got = maximum_path(value.cuda(), mask.cuda()).cpu()
    ref0 = mas_oracle.maximum_path(value[:1].numpy(), mask[:1].numpy())
    assert np.array_equal(got[0].numpy(), ref0[0])                 # the healthy item is the reference's path
    assert bool(((got == 0) | (got == 1)).all()) and bool((got <= mask).all())
    assert got[1].sum() <= t_y and got[2].sum() <= t_y
