"""Greedy decoding on the device (beam_width == 1 of Models/decoder.py:63-130)."""
import ctypes as C

import torch

from .. import _lib, ops
from .._lib import AttDecFwdArgs, check

NC = 10


def greedy_decode(dec, hbatch, lengths, max_len, eos_id):
    """Returns (result, raw): ``result`` is what the reference's decode() returns
    (the sequence up to and including the first eos, or [] if none within
    ``max_len`` steps); ``raw`` is the full argmax sequence."""
    lib = _lib.load()
    hbatch = hbatch.contiguous().float()
    B, T, E = hbatch.shape
    dev = hbatch.device
    f32 = dict(device=dev, dtype=torch.float32)
    (W_se, W_he, b_he, w_ee, W_fe, conv_w, W_sy, W_gy, b_gy, W_yy, b_yy, Emb, W_ss, W_gs, b_gs) = \
        [p.detach() for p in dec._params()]
    Hd = W_sy.shape[0]
    A, V, L = 2 * Hd, W_yy.shape[0], int(max_len)
    from ..utils.utils import lengths_to_device
    lengths_dev = lengths_to_device(lengths, dev)
    PH = torch.empty(B, T, A, **f32)
    ops.linear_nt(hbatch.view(B * T, E), W_he, PH.view(B * T, A), bias1=b_he)
    tokens = torch.zeros(1, L, device=dev, dtype=torch.int64)
    t = dict(hbatch=hbatch, PH=PH, lengths=lengths_dev, targets=tokens, W_se=W_se, w_ee=w_ee, W_fe=W_fe,
             conv_w=conv_w, Emb=Emb, W_ss=W_ss, W_gs=W_gs, b_gs=b_gs,
             S=torch.empty(L + 1, B, Hd, **f32), C=torch.empty(L + 1, B, Hd, **f32),
             alpha=torch.empty(L, B, T, **f32), sproj=torch.empty(L, B, A, **f32),
             g=torch.empty(L, B, E, **f32), gates=torch.empty(L, B, 4 * Hd, **f32),
             Fc=torch.empty(L, B, T, NC, **f32),
             ws=torch.empty(int(lib.asr_attdec_ws_floats(B, T, E, Hd, L)), **f32))
    a = AttDecFwdArgs()
    a.B, a.T, a.E, a.Hd, a.L, a.V = B, T, E, Hd, L, V
    for k, v in t.items():
        setattr(a, k, v.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.asr_attdec_greedy(C.byref(a), W_sy.data_ptr(), W_gy.data_ptr(), b_gy.data_ptr(), W_yy.data_ptr(),
                                b_yy.data_ptr(), tokens.data_ptr(), st))
    raw = tokens[0].tolist()            # one device->host copy for the whole utterance
    if eos_id in raw:
        return raw[:raw.index(eos_id) + 1], raw
    return [], raw
