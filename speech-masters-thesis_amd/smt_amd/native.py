"""ctypes loader for libsmt_hip.so -- the C ABI declared in include/smt_hip.h."""
import contextlib
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMT_HIP_LIB: an alternative build of the same ABI (tools/ablate_*.sh link their -D ablation builds to libsmt_hip_abl.so)
LIB_PATH = os.environ.get("SMT_HIP_LIB") or os.path.join(_HERE, "libsmt_hip.so")
ABI_VERSION = 3

_lib = None
_lock = threading.Lock()

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_u32 = ctypes.c_uint32
c_ptr = ctypes.c_void_p
c_size = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/smt_hip.h one to one
_SIGNATURES = {
    "smt_last_error": (ctypes.c_char_p, []),
    "smt_abi_version": (c_int, []),
    "smt_vq_forward_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "smt_vq_prep_bytes": (c_size, [c_int, c_int]),
    "smt_vq_prepare": (c_int, [c_ptr, c_int, c_int, c_ptr, c_size, c_ptr]),
    "smt_vq_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_size,
                               c_ptr]),
    "smt_vq_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr]),
    "smt_vq_ema_accumulate_workspace_bytes": (c_size, [c_i64, c_int, c_int]),
    "smt_vq_ema_accumulate": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "smt_vq_ema_apply": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_int, c_int, c_ptr, c_ptr, c_size,
                                 c_ptr]),
    "smt_pack_weight": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_ptr, c_int, c_ptr]),
    "smt_pack_weights_batched": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr]),
    "smt_conv1d_ntc": (c_int, [c_ptr, c_ptr]),
    "smt_conv1d_kernel_name": (ctypes.c_char_p, [c_ptr]),
    "smt_conv1d_wgrad_workspace_bytes": (c_size, [c_ptr]),
    "smt_conv1d_wgrad_kernel_name": (ctypes.c_char_p, [c_ptr]),
    "smt_conv1d_wgrad": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    "smt_wgrad_reduce_defer": (c_int, [c_int, c_ptr]),
    "smt_conv1x1_bwd_workspace_bytes": (c_size, [c_ptr]),
    "smt_conv1x1_bwd": (c_int, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "smt_conv_gate_bwd_workspace_bytes": (c_size, [c_int, c_int]),
    "smt_conv_gate_bwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_int,
                                  c_int, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "smt_conv_k1_bwd_workspace_bytes": (c_size, [c_int, c_int]),
    "smt_conv_k1_bwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_i64,
                                c_int, c_ptr, c_int, c_int, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "smt_convt4s2": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "smt_conv4s2": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "smt_conv_k3gate_fwd": (c_int, [c_ptr, c_i64, c_int, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int,
                                    c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr]),
    "smt_gate_mix_fwd": (c_int, [c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_gate_mix_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_conv_in_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_ptr]),
    "smt_conv_in_wgrad_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "smt_conv_in_wgrad": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_ptr, c_size, c_ptr]),
    "smt_conv_out_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_conv_out_bwd_workspace_bytes": (c_size, [c_int, c_int, c_int]),
    "smt_maximum_path": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_f32, c_ptr, c_ptr]),
    "smt_recon_loss_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "smt_recon_loss_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr]),
    "smt_stft_num_frames": (c_int, [c_int, c_int, c_int]),
    "smt_stft_magnitude": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_melspec": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_stft_loss_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_stft_loss_bwd_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "smt_stft_loss_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int,
                                  c_ptr, c_size, c_ptr]),
    "smt_stft_inverse": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_fft_selftest": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "smt_conv_out_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr,
                                 c_size, c_ptr]),
    "smt_lm_make_keys": (c_int, [c_ptr, c_ptr, c_int, c_ptr]),
    "smt_lm_embed_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_f32, c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_lm_embed_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_f32, c_u32, c_ptr, c_u32, c_f32, c_i64,
                                 c_ptr]),
    "smt_lm_attention_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_lm_attention_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_u32, c_ptr, c_u32, c_f32,
                                     c_ptr]),
    "smt_lm_add_ln_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_f32, c_u32, c_ptr, c_u32, c_f32,
                                  c_ptr]),
    "smt_lm_add_ln_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "smt_lm_add_ln_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_u32, c_ptr,
                                  c_u32, c_f32, c_ptr, c_size, c_ptr]),
    "smt_lm_bias_relu_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_lm_bias_relu_bwd_workspace_bytes": (c_size, [c_i64, c_int]),
    "smt_lm_bias_relu_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_u32, c_ptr, c_u32, c_f32, c_ptr, c_size, c_ptr]),
    "smt_lm_ce_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "smt_lm_ce_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_ptr]),
    "smt_glow_reduce_workspace_bytes": (c_size, [c_i64, c_int]),
    "smt_glow_actnorm_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_glow_actnorm_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_size, c_ptr]),
    "smt_glow_invconv": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_glow_invconv_wgrad": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_size, c_ptr]),
    "smt_glow_gate_fwd": (c_int, [c_ptr, c_ptr, c_i64, c_int, c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_glow_gate_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_glow_dropout": (c_int, [c_ptr, c_ptr, c_i64, c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_glow_coupling_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_size, c_ptr]),
    "smt_glow_coupling_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_glow_attention_fwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int,
                                       c_u32, c_ptr, c_u32, c_f32, c_ptr]),
    "smt_glow_attention_bwd_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "smt_glow_attention_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int,
                                       c_int, c_int, c_int, c_u32, c_ptr, c_u32, c_f32, c_ptr, c_size, c_ptr]),
    "smt_glow_prior_logp": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_glow_align_index": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr]),
    "smt_glow_align_gather": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_glow_align_scatter": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "smt_glow_mle_workspace_bytes": (c_size, [c_i64]),
    "smt_glow_mle_sums": (c_int, [c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_size, c_ptr]),
    "smt_glow_mle_bwd": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr]),
    "smt_glow_length_loss": (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr]),
}


class NativeLibraryError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load (once) and return the ctypes handle; fail loudly if it is missing."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise NativeLibraryError(
                        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU/eager fallback for the hot path)")
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(handle, name)  # AttributeError if the .so is stale
                    fn.restype = res
                    fn.argtypes = args
                if handle.smt_abi_version() != ABI_VERSION:
                    raise NativeLibraryError("libsmt_hip.so ABI version mismatch; rebuild")
                _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().smt_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")


def ptr(t):
    """Device pointer of a tensor (or NULL for None) as a void*."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "native ops need contiguous device tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Workspace:
    """Grow-only per-device scratch buffer handed to the kernels (the library never allocates).

    Inside ``with workspace.arena(device):`` every ``get`` returns a region of its OWN (256-byte aligned slices of one
    buffer) that stays untouched until the block ends -- what the deferred slab reductions need (smt_wgrad_reduce_defer)."""

    def __init__(self):
        self._buf = {}
        self._arena = None          # (key, [chunks], offset) while an arena is open
        self._arena_buf = {}        # key -> buffer kept between arenas (sized for the largest one seen)

    @contextlib.contextmanager
    def arena(self, device):
        key = (device.type, device.index)
        assert self._arena is None, "workspace arenas do not nest"
        first = self._arena_buf.get(key)
        self._arena = [key, [first] if first is not None else [], 0, 0]     # key, chunks, offset in the last chunk, total bytes
        try:
            yield self
        finally:
            _, chunks, _, total = self._arena
            self._arena = None
            if len(chunks) > 1 or first is None:      # outgrown: one buffer of the full size for the next time
                self._arena_buf[key] = torch.empty(max(int(total * 1.25), 1 << 20), dtype=torch.uint8, device=device)

    def _arena_get(self, nbytes, device):
        key, chunks, off, total = self._arena
        assert key == (device.type, device.index)
        need = (int(nbytes) + 255) // 256 * 256
        if not chunks or off + need > chunks[-1].numel():
            chunks.append(torch.empty(max(need, 64 << 20), dtype=torch.uint8, device=device))
            off = 0
        self._arena[2], self._arena[3] = off + need, total + need
        return chunks[-1][off:off + need]

    def get(self, nbytes, device):
        if self._arena is not None:
            return self._arena_get(nbytes, device)
        key = (device.type, device.index)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


workspace = Workspace()
