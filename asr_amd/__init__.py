"""asr_amd -- MI355X-native AttModel training hot path (drop-in for syoamakase/ASR).

Mirrors the reference's package layout for the path it replaces:
``asr_amd.Models.AttModel.AttModel`` (Models/AttModel.py),
``asr_amd.Loss.label_smoothing.label_smoothing_loss`` (Loss/label_smoothing.py),
``asr_amd.utils.utils.frame_stacking`` (utils/utils.py:47-65).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
