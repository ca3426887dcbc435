"""Workspaces of the persistent recurrence kernels and their sticky error word.

A persistent kernel (csrc/blstm.hip) hands data between workgroups inside one launch; every wait is
bounded, and a wait that runs out sets word ``ERR_WORD`` of the workspace and lets the grid drain --
the layer's output is then garbage.  The reference has no such failure mode; what it does have is
"NaN loss -> exit" (train.py:66-68).  ``poison(loss)`` folds the word into the loss on the device
(no host sync, one 1-thread launch per workspace), so the training loop's own ``isnan(loss)`` ends
the job.  The backward of a BLSTM layer folds the word into its bias gradient right behind the
launch (``poison_from``, Models/encoder.py), as the decoder does with dPH: a time-out in a backward
pass turns that step's gradients into NaN instead of applying one update of garbage, and the word
being sticky, the next loss is NaN as well.

One workspace per (device, B, H, stream): launches on one stream are serialised, so all layers and
steps on that stream share it; two encoders running on two streams of one process get two.
"""
import ctypes as C

import torch

ERR_WORD = 1024
_WS = {}
_USED = set()      # keys of the workspaces handed out since the last poison(): only those are checked


def workspace(lib, B, H, dev):
    key = (dev.index, B, H, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev, dtype=torch.float32)
        _WS[key] = ws
    _USED.add(key)
    return ws


def poison_from(ws, value):
    """``value[0]`` (float32, device) becomes NaN if workspace ``ws`` carries a timed-out wait.
    Asynchronous, on the current stream (which must be ordered behind the launch that used ``ws``)."""
    from . import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream(value.device).cuda_stream)
    _lib.check(lib.asr_poison_if_flag(ws.data_ptr() + 4 * ERR_WORD, value.data_ptr(), st))


def error_flags():
    """True if any persistent kernel reported a timed-out wait (device -> host copy: tests, end of a run)."""
    return any(bool(ws.view(torch.int32)[ERR_WORD].item()) for ws in _WS.values())


def poison(value):
    """``value`` (a 0-dim float32 device tensor) becomes NaN if a persistent kernel that ran on its
    device since the last call has reported a timed-out wait (the word is sticky and the job ends on
    the first NaN, so workspaces that were not used since need no second look: the launch count stays
    at the shapes of THIS step however many batch sizes the process has seen).  Asynchronous, on the
    current stream."""
    if not _USED:
        return
    for key in [k for k in _USED if k[0] == value.device.index]:
        _USED.discard(key)
        poison_from(_WS[key], value)
