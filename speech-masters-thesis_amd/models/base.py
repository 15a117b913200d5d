"""Model API of the reference (models/base.py:6-55): four abstract families, each
mapping the 7-slot batch ``(token, token_len, spect, spect_len, audio, audio_len,
speaker)`` onto ``forward`` and returning ``(loss_dict, metrics_dict)``.

``isinstance`` against these classes drives the dataset flag surgery in
utils/commons.get_model and the validation artefact type in train.py, so the
class names and the slot mapping are part of the drop-in contract.
"""
import torch.nn as nn

_SLOTS = ("token", "token_len", "spect", "spect_len", "audio", "audio_len", "speaker")


class _SlotModel(nn.Module):
    inputs = ()        # batch slots passed positionally to forward
    target = None      # slot stored back as loss_dict["y"]
    squeeze_target = False

    def supervised_step(self, batch):
        named = dict(zip(_SLOTS, batch))
        loss_dict, metrics_dict = self(*[named[s] for s in self.inputs], speaker=named["speaker"])
        y = named[self.target]
        loss_dict["y"] = y.squeeze(1) if self.squeeze_target else y
        return loss_dict, metrics_dict

    def forward(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement forward")


class TokenToWaveformModel(_SlotModel):
    inputs, target, squeeze_target = ("token", "token_len", "audio", "audio_len"), "audio", True


class WaveformReconstructionModel(_SlotModel):
    inputs, target, squeeze_target = ("audio", "audio_len"), "audio", True


class TokenToSpectrogramModel(_SlotModel):
    inputs, target = ("token", "token_len", "spect", "spect_len"), "spect"


class SpectrogramReconstructionModel(_SlotModel):
    inputs, target = ("spect", "spect_len"), "spect"
