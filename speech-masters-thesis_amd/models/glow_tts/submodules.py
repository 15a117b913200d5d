"""Only the helper the VQ-VAE path imports from the GlowTTS package (reference
models/glow_tts/submodules.py:18-25, used at models/vqvae/vqvae.py:99)."""
import torch


def sequence_mask(length, max_length=None):
    if max_length is None:
        max_length = int(length.max())
    steps = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return steps[None, :] < length[:, None]
