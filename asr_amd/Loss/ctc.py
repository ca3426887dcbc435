"""CTC loss on the HIP kernels -- drop-in for train.py:56-57 of the reference:

    predict_ts = F.log_softmax(predict_ts, dim=2).transpose(0, 1)
    loss = F.ctc_loss(predict_ts, text, mel_lengths, text_lengths, blank=0)

becomes ``loss = ctc_loss(predict_ts, text, mel_lengths, text_lengths, blank=0)`` with
``predict_ts`` the (B,T,V) output of CTCModel.forward.  One library call
(asr_ctc_loss_fwd_bwd) computes the loss and d loss / d logits; the (T,B,V) log-probability
tensor and its transpose are never materialised.
"""
import ctypes as C

import torch

from .. import _lib, persist
from .._lib import check
from ..utils.utils import lengths_to_device


class _CTCFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, in_len, tgt_len, blank):
        lib = _lib.load()
        logits = logits.contiguous()
        B, T, V = logits.shape
        L = targets.shape[1]
        dev = logits.device
        f32 = dict(device=dev, dtype=torch.float32)
        ws = torch.empty(int(lib.asr_ctc_ws_floats(B, T, L)), **f32)
        nll = torch.empty(B, **f32)
        loss = torch.empty((), **f32)
        dlogits = torch.empty_like(logits)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.asr_ctc_loss_fwd_bwd(logits.data_ptr(), targets.data_ptr(), in_len.data_ptr(), tgt_len.data_ptr(),
                                       B, T, V, L, int(blank), ws.data_ptr(), nll.data_ptr(), loss.data_ptr(),
                                       dlogits.data_ptr(), st))
        persist.poison(loss)      # NaN if a persistent recurrence kernel reported a timed-out wait
        ctx.dlogits = dlogits if ctx.needs_input_grad[0] else None
        ctx.mark_non_differentiable(nll)
        return loss, nll

    @staticmethod
    def backward(ctx, grad_loss, _grad_nll):
        d = ctx.dlogits
        ctx.dlogits = None
        if d is None:
            return (None,) * 5
        return d * grad_loss, None, None, None, None


def ctc_loss(predict_ts, text, mel_lengths, text_lengths, blank=0, return_nll=False):
    """predict_ts (B,T,V) logits on the GPU, text (B,L) int64, mel_lengths (B,) frames per
    utterance (CPU or device, as train.py passes them), text_lengths (B,).  Returns the scalar
    loss of F.ctc_loss(reduction='mean'); with return_nll also the (B,) per-utterance nll."""
    if not predict_ts.is_cuda:
        raise RuntimeError("asr_amd.ctc_loss: logits must be on the GPU; no CPU fallback exists")
    if predict_ts.dim() != 3 or text.dim() != 2 or text.shape[0] != predict_ts.shape[0]:
        raise RuntimeError("asr_amd.ctc_loss: expected logits (B,T,V) and targets (B,L)")
    dev = predict_ts.device
    targets = text.to(device=dev, dtype=torch.int64).contiguous()
    in_len = lengths_to_device(mel_lengths, dev).contiguous()
    tgt_len = lengths_to_device(text_lengths, dev).contiguous()
    loss, nll = _CTCFunction.apply(predict_ts.float(), targets, in_len, tgt_len, blank)
    return (loss, nll) if return_nll else loss
