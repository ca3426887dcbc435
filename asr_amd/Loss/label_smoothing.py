"""Fused label-smoothed cross-entropy -- drop-in for Loss/label_smoothing.py:14-28.

One HIP kernel (asr_ls_ce_fwd_bwd) reads the logits once and produces both the
loss and d loss / d logits; the reference's log_softmax tensor, dense one-hot
tensor and per-utterance python loop (with a host sync each) disappear.
"""
import ctypes as C

import torch

from .. import _lib, persist
from .._lib import check


class _LSCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, text_lengths, T_norm, B_norm, eps):
        lib = _lib.load()
        logits = logits.contiguous()
        B, L, V = logits.shape
        dev = logits.device
        row_loss = torch.empty(B * L, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        dlogits = torch.empty_like(logits) if need else None
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.asr_ls_ce_fwd_bwd(logits.data_ptr(), target.data_ptr(), text_lengths.data_ptr(), B, L, V,
                                    C.c_float(eps), int(bool(T_norm)), int(bool(B_norm)), row_loss.data_ptr(),
                                    loss.data_ptr(), 0 if dlogits is None else dlogits.data_ptr(), st))
        persist.poison(loss)      # NaN if a persistent recurrence kernel reported a timed-out wait
        ctx.dlogits = dlogits
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        d = ctx.dlogits
        ctx.dlogits = None
        if d is None:
            return (None,) * 6
        return d * grad_out, None, None, None, None, None


def label_smoothing_loss(predicted_label, target, text_lengths, T_norm=True, B_norm=False, eps=0.1):
    """Same signature and meaning as the reference.  ``target`` must be contiguous
    (the reference's ``.view(-1, 1)`` demands it too)."""
    if not target.is_contiguous():
        raise RuntimeError("view size is not compatible with input tensor's size and stride "
                           "(target must be contiguous, as in Loss/label_smoothing.py:18)")
    if not predicted_label.is_cuda:
        raise RuntimeError("asr_amd.label_smoothing_loss: logits must be on the GPU; no CPU fallback exists")
    dev = predicted_label.device
    target = target.to(device=dev, dtype=torch.int64)
    text_lengths = text_lengths.to(device=dev, dtype=torch.int64).contiguous()
    return _LSCEFunction.apply(predicted_label.float(), target, text_lengths, T_norm, B_norm, float(eps))
