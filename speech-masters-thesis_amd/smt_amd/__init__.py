"""Python binding layer over libsmt_hip.so (hand-written gfx950 kernels).

PyTorch is used for device memory, streams and autograd plumbing only; the
arithmetic of every op in this package runs in the HIP library.  There is no
CPU or eager fallback: a missing/unloadable library raises at first use.
"""
from . import native  # noqa: F401
