"""Beam search on the device kernels -- Models/decoder.py:63-130 with beam_width > 1 (batch 1,
no LM fusion).

The live hypotheses are the batch dimension of the same per-step kernels the training forward
uses (asr_attdec_step_att / asr_attdec_step_cell); scoring, the per-hypothesis top-k and the
stable sort over all candidates are torch ops on the device; one small device->host copy per
step (the survivors' parents, tokens and scores) replaces the reference's per-candidate
``.item()`` calls.  Candidate order, score arithmetic (float32 ``old + new``) and the stable
descending sort follow the reference, so the surviving sequences are the same.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from .. import _lib, ops
from .._lib import AttDecFwdArgs, check
from ..utils.utils import lengths_to_device

NC = 10


def beam_decode(dec, hbatch, lengths, beam_width, max_len, eos_id, score_func="log_softmax"):
    """Returns (result, trace): ``result`` is what the reference's decode() returns (the best
    hypothesis as soon as it ends in eos, else [] after ``max_len`` steps); trace[k] = the
    survivors of step k as (tokens, score)."""
    if hbatch.shape[0] != 1:
        raise ValueError("beam search is batch-1, as in the reference")
    if score_func not in ("log_softmax", "softmax"):
        raise ValueError(f"unknown score_func {score_func!r}")
    lib = _lib.load()
    W = int(beam_width)
    hbatch = hbatch.contiguous().float()
    _, T, E = hbatch.shape
    dev = hbatch.device
    f32 = dict(device=dev, dtype=torch.float32)
    (W_se, W_he, b_he, w_ee, W_fe, conv_w, W_sy, W_gy, b_gy, W_yy, b_yy, Emb, W_ss, W_gs, b_gs) = \
        [p.detach() for p in dec._params()]
    Hd = W_sy.shape[0]
    A, V = 2 * Hd, W_yy.shape[0]
    # hoisted L_he(hbatch); every hypothesis attends over the same utterance
    PH1 = torch.empty(1, T, A, **f32)
    ops.linear_nt(hbatch.view(T, E), W_he, PH1.view(T, A), bias1=b_he)
    hb = hbatch.expand(W, T, E).contiguous()
    PH = PH1.expand(W, T, A).contiguous()
    len_dev = lengths_to_device(lengths, dev).expand(W).contiguous()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    seqs = [[]]
    scores = torch.zeros(1, **f32)
    S = torch.zeros(1, Hd, **f32)
    Cc = torch.zeros(1, Hd, **f32)
    alpha_prev = None
    trace = []
    for _ in range(int(max_len)):
        n = len(seqs)
        S_io = torch.empty(2, n, Hd, **f32)
        C_io = torch.empty(2, n, Hd, **f32)
        S_io[0].copy_(S)
        C_io[0].copy_(Cc)
        alpha = torch.empty(n, T, **f32)
        g = torch.empty(n, E, **f32)
        t = dict(hbatch=hb, PH=PH, lengths=len_dev, targets=torch.zeros(n, dtype=torch.int64, device=dev),
                 W_se=W_se, w_ee=w_ee, W_fe=W_fe, conv_w=conv_w, Emb=Emb, W_ss=W_ss, W_gs=W_gs, b_gs=b_gs,
                 S=S_io, C=C_io, alpha=alpha, sproj=torch.empty(n, A, **f32), g=g,
                 gates=torch.empty(n, 4 * Hd, **f32), Fc=torch.empty(n, T, NC, **f32),
                 ws=torch.empty(int(lib.asr_attdec_ws_floats(n, T, E, Hd, 1)), **f32))
        a = AttDecFwdArgs()
        a.B, a.T, a.E, a.Hd, a.L, a.V = n, T, E, Hd, 1, V
        for k, v in t.items():
            setattr(a, k, v.data_ptr())
        check(lib.asr_attdec_step_att(C.byref(a), 0 if alpha_prev is None else alpha_prev.data_ptr(), st))
        # y = L_yy tanh(L_gy g + L_sy s)   (Models/decoder.py:89)
        U = torch.empty(n, Hd, **f32)
        ops.linear_nt(g, W_gy, U, bias1=b_gy)
        ops.linear_nt(S_io[0], W_sy, U, accumulate=True, act=1)
        Y = torch.empty(n, V, **f32)
        ops.linear_nt(U, W_yy, Y, bias1=b_yy)
        y = F.log_softmax(Y, dim=1) if score_func == "log_softmax" else F.softmax(Y, dim=1)
        k = min(W, V)
        vals, idx = torch.topk(y, k, dim=1)                     # the repeated argmax of :109-116
        cand = (scores.view(n, 1) + vals).reshape(-1)           # hypothesis-major, like the append order
        order = torch.sort(cand, descending=True, stable=True).indices[:W]
        parent = order // k
        tok = idx.reshape(-1)[order]
        host = torch.stack([parent.double(), tok.double(), cand[order].double()]).tolist()   # one copy per step
        parents = [int(v) for v in host[0]]
        toks = [int(v) for v in host[1]]
        m = len(parents)
        # the survivors' cell update (Models/decoder.py:118-119) from their parents' state and context
        S2 = torch.empty(2, m, Hd, **f32)
        C2 = torch.empty(2, m, Hd, **f32)
        S2[0].copy_(S_io[0].index_select(0, parent))
        C2[0].copy_(C_io[0].index_select(0, parent))
        g2 = g.index_select(0, parent).contiguous()
        t.update(S=S2, C=C2, g=g2, targets=tok.contiguous(), gates=torch.empty(m, 4 * Hd, **f32))
        a.B = m
        for k2 in ("S", "C", "g", "targets", "gates"):
            setattr(a, k2, t[k2].data_ptr())
        check(lib.asr_attdec_step_cell(C.byref(a), st))
        S, Cc = S2[1], C2[1]
        alpha_prev = alpha.index_select(0, parent).contiguous()
        scores = cand[order].contiguous()
        seqs = [seqs[p] + [tk] for p, tk in zip(parents, toks)]
        trace.append([(list(sq), float(sc)) for sq, sc in zip(seqs, host[2])])
        if seqs[0][-1] == eos_id:
            return list(seqs[0]), trace
    return [], trace
