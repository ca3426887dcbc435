"""Beam search on the device kernels (batch 1): ``beam_decode`` = Models/decoder.py:63-130
with beam_width > 1 (or any width with a language model), ``beam_decode_v2`` = the second search,
Models/decoder.py:132-267 (decode_v2).  Shallow fusion (:93-96, :180-183): ``model_lm`` is any callable
``tokens (n, len) int64 on the device -> scores (n, len, V)``; the live hypotheses go through it as ONE batch per
step (they all have the same length) where the reference's decode() calls it once per hypothesis.

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


class _Stepper:
    """One utterance's per-step kernel calls with the hypotheses as batch rows: ``att`` = attention
    step + output layer (Models/decoder.py:85-89 / :172-176), ``cell`` = the survivors' LSTM cell
    update (:118-119 / :249-250)."""

    def __init__(self, dec, hbatch, lengths, rows):
        self.lib = _lib.load()
        hbatch = hbatch.contiguous().float()
        _, T, E = hbatch.shape
        dev = hbatch.device
        self.f32 = f32 = dict(device=dev, dtype=torch.float32)
        (self.W_se, W_he, b_he, self.w_ee, self.W_fe, self.conv_w, self.W_sy, self.W_gy, self.b_gy, self.W_yy,
         self.b_yy, self.Emb, self.W_ss, self.W_gs, self.b_gs) = [p.detach() for p in dec._params()]
        self.Hd = Hd = self.W_sy.shape[0]
        self.A, self.V, self.T, self.E, self.dev = 2 * Hd, self.W_yy.shape[0], T, E, dev
        # hoisted L_he(hbatch); every hypothesis attends over the same utterance
        PH1 = torch.empty(1, T, self.A, **f32)
        ops.linear_nt(hbatch.view(T, E), W_he, PH1.view(T, self.A), bias1=b_he)
        self.hb = hbatch.expand(rows, T, E).contiguous()
        self.PH = PH1.expand(rows, T, self.A).contiguous()
        self.len_dev = lengths_to_device(lengths, dev).expand(rows).contiguous()
        self.st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _args(self, n, t):
        a = AttDecFwdArgs()
        a.B, a.T, a.E, a.Hd, a.L, a.V = n, self.T, self.E, self.Hd, 1, self.V
        for k, v in t.items():
            setattr(a, k, v.data_ptr())
        return a

    def att(self, S, Cc, alpha_prev):
        """S, Cc (n,Hd), alpha_prev (n,T) or None (zeros) -> g (n,E), alpha (n,T), logits (n,V)."""
        n, Hd, T, f32 = S.shape[0], self.Hd, self.T, self.f32
        S_io = torch.empty(2, n, Hd, **f32)
        C_io = torch.empty(2, n, Hd, **f32)
        S_io[0].copy_(S)
        C_io[0].copy_(Cc)
        alpha = torch.empty(n, T, **f32)
        g = torch.empty(n, self.E, **f32)
        self.t = t = dict(
            hbatch=self.hb, PH=self.PH, lengths=self.len_dev, targets=torch.zeros(n, dtype=torch.int64, device=self.dev),
            W_se=self.W_se, w_ee=self.w_ee, W_fe=self.W_fe, conv_w=self.conv_w, Emb=self.Emb, W_ss=self.W_ss,
            W_gs=self.W_gs, b_gs=self.b_gs, S=S_io, C=C_io, alpha=alpha, sproj=torch.empty(n, self.A, **f32), g=g,
            gates=torch.empty(n, 4 * Hd, **f32), Fc=torch.empty(n, T, NC, **f32),
            ws=torch.empty(int(self.lib.asr_attdec_ws_floats(n, T, self.E, Hd, 1)), **f32))
        a = self._args(n, t)
        check(self.lib.asr_attdec_step_att(C.byref(a), 0 if alpha_prev is None else alpha_prev.data_ptr(), self.st))
        # y = L_yy tanh(L_gy g + L_sy s)
        U = torch.empty(n, Hd, **f32)
        ops.linear_nt(g, self.W_gy, U, bias1=self.b_gy)
        ops.linear_nt(S_io[0], self.W_sy, U, accumulate=True, act=1)
        Y = torch.empty(n, self.V, **f32)
        ops.linear_nt(U, self.W_yy, Y, bias1=self.b_yy)
        return g, alpha, Y

    def cell(self, S, Cc, g, parent, tok):
        """State after consuming ``tok`` (m,) from the parents' (S, Cc, g) rows ``parent`` (m,)."""
        m, Hd, f32 = parent.shape[0], self.Hd, self.f32
        S2 = torch.empty(2, m, Hd, **f32)
        C2 = torch.empty(2, m, Hd, **f32)
        S2[0].copy_(S.index_select(0, parent))
        C2[0].copy_(Cc.index_select(0, parent))
        t = dict(self.t)
        t.update(S=S2, C=C2, g=g.index_select(0, parent).contiguous(), targets=tok.contiguous(),
                 gates=torch.empty(m, 4 * Hd, **f32))
        a = self._args(m, t)
        check(self.lib.asr_attdec_step_cell(C.byref(a), self.st))
        return S2[1], C2[1]


def _fuse_lm(y, seqs, model_lm, lm_weight, score_func, dev):
    """y + lm_weight * log_softmax(model_lm(seqs)[:, -1]) (Models/decoder.py:94-96, :181-183); the reference's
    'softmax' branch with an LM (:101-106, :188-193) reads scores it never assigned, so that combination is refused."""
    if score_func != "log_softmax":
        raise ValueError("shallow fusion needs score_func == 'log_softmax' (the reference's softmax branch is broken)")
    lm_score = model_lm(torch.tensor(seqs, dtype=torch.int64, device=dev))[:, -1, :]
    if lm_score.shape != y.shape:
        raise RuntimeError(f"model_lm returned scores of shape {tuple(lm_score.shape)}, expected {tuple(y.shape)}")
    return y + float(lm_weight) * F.log_softmax(lm_score.to(y.dtype), dim=1)


def beam_decode(dec, hbatch, lengths, beam_width, max_len, eos_id, score_func="log_softmax", model_lm=None,
                lm_weight=0.0):
    """Returns (result, trace): ``result`` is what the reference's decode() returns (the best
    hypothesis as soon as it ends in eos, else [] after ``max_len`` steps); trace[k] = the
    survivors of step k as (tokens, score)."""
    if hbatch.shape[0] != 1:
        raise ValueError("beam search is batch-1, as in the reference")
    if score_func not in ("log_softmax", "softmax"):
        raise ValueError(f"unknown score_func {score_func!r}")
    W = int(beam_width)
    k = _Stepper(dec, hbatch, lengths, W)
    f32 = k.f32

    seqs = [[]]
    scores = torch.zeros(1, **f32)
    S = torch.zeros(1, k.Hd, **f32)
    Cc = torch.zeros(1, k.Hd, **f32)
    alpha_prev = None
    trace = []
    for _ in range(int(max_len)):
        n = len(seqs)
        g, alpha, Y = k.att(S, Cc, alpha_prev)
        y = F.log_softmax(Y, dim=1) if score_func == "log_softmax" else F.softmax(Y, dim=1)
        if model_lm is not None and (score_func != "log_softmax" or len(seqs[0]) > 0):     # :93: not on the first step
            y = _fuse_lm(y, seqs, model_lm, lm_weight, score_func, k.dev)
        kk = min(W, k.V)
        vals, idx = torch.topk(y, kk, dim=1)                    # the repeated argmax of :109-116
        cand = (scores.view(n, 1) + vals).reshape(-1)           # hypothesis-major, like the append order
        order = torch.sort(cand, descending=True, stable=True).indices[:W]
        parent = order // kk
        tok = idx.reshape(-1)[order]
        host = torch.stack([parent.double(), tok.double(), cand[order].double()]).tolist()   # one copy per step
        parents = [int(v) for v in host[0]]
        toks = [int(v) for v in host[1]]
        # the survivors' cell update (Models/decoder.py:118-119) from their parents' state and context
        S, Cc = k.cell(S, Cc, g, parent, tok)
        alpha_prev = alpha.index_select(0, parent).contiguous()
        scores = cand[order].contiguous()
        seqs = [seqs[p] + [tk] for p, tk in zip(parents, toks)]
        trace.append([(list(sq), float(sc)) for sq, sc in zip(seqs, host[2])])
        if seqs[0][-1] == eos_id:
            return list(seqs[0]), trace
    return [], trace


def beam_decode_v2(dec, hbatch, lengths, beam_width, max_len, eos_id, score_func="log_softmax", model_lm=None,
                   lm_weight=0.0):
    """Models/decoder.py:132-267 (decode_v2) on the device kernels.

    The ``beam_width`` live hypotheses are the batch rows of one attention step; each proposes its
    ``beam_width`` best tokens, scored ``old + (log-)probability + 1`` in float32 (:195-196).
    Step 0 takes row 0's proposals as the first beam (:200-210).  Afterwards the best
    ``2 * beam_width`` proposals are walked in score order (:212-255): one ending in eos goes to
    the next result slot, any other fills the next beam slot with its parent's state; the search
    ends when ``beam_width`` results exist or after ``max_len`` steps.  The answer is the result
    slot with the highest score, empty slots counting 0 with length 0 (:256-258) -- so [] when
    every finished hypothesis scores below 0.  One device->host copy per step (scores, parents,
    tokens of the 2 * beam_width proposals); the walk itself is host logic as in the reference.

    Returns (result, trace): trace[k] = (live survivors of step k as (tokens, score), results so
    far as (tokens, score)).  Raises RuntimeError, as the reference does (:219,241), when the
    encoder output width is not 2 * num_hidden_nodes_decoder, and for widths the reference's
    ``topk`` calls reject (beam_width < 2 or > num_classes)."""
    if hbatch.shape[0] != 1:
        raise ValueError("beam search is batch-1, as in the reference")
    if score_func not in ("log_softmax", "softmax"):
        raise ValueError(f"unknown score_func {score_func!r}")
    W = int(beam_width)
    if hbatch.shape[2] != 2 * dec.num_decoder_hidden_nodes:
        raise RuntimeError("decode_v2: encoder output width must be 2 * num_hidden_nodes_decoder (Models/decoder.py:219)")
    if W < 2 or W > dec.num_classes:
        raise RuntimeError("decode_v2: beam_width must be in [2, num_classes] (topk at Models/decoder.py:195,212)")
    k = _Stepper(dec, hbatch, lengths, W)
    f32, dev = k.f32, k.dev
    seqs = [[] for _ in range(W)]
    score = torch.zeros(W, **f32)
    S = torch.zeros(W, k.Hd, **f32)
    Cc = torch.zeros(W, k.Hd, **f32)
    alpha_prev = None
    done, trace = [], []
    for step in range(int(max_len)):
        g, alpha, Y = k.att(S, Cc, alpha_prev)
        y = F.log_softmax(Y, dim=1) if score_func == "log_softmax" else F.softmax(Y, dim=1)
        if model_lm is not None and (score_func != "log_softmax" or step > 0):              # :180: from the second step on
            y = _fuse_lm(y, seqs, model_lm, lm_weight, score_func, dev)
        best, idx = torch.topk(y, W, dim=1)
        cand = score.view(W, 1) + best + 1
        if step == 0:
            # all rows are identical here; row i keeps its own (zero) state
            first = torch.stack([idx[0].double(), cand[0].double()]).tolist()
            parents, toks, host_scores = list(range(W)), [int(v) for v in first[0]], first[1]
            parent, tok, score = torch.arange(W, device=dev), idx[0].contiguous(), cand[0].contiguous()
        else:
            k_scores, k_ix = torch.topk(cand.reshape(-1), 2 * W)
            host = torch.stack([k_scores.double(), k_ix.double(), idx.reshape(-1)[k_ix].double()]).tolist()
            parents, toks, kept = [], [], []
            for pos, (sc, ix, tk) in enumerate(zip(*host)):
                h, tk = int(ix) // W, int(tk)
                if tk == eos_id:
                    done.append((seqs[h] + [tk], sc))
                else:
                    parents.append(h)
                    toks.append(tk)
                    kept.append(pos)
                if len(parents) == W or len(done) >= W:
                    break
            if len(done) >= W:
                trace.append(([], list(done)))
                break
            sel = torch.tensor([parents, toks, kept], dtype=torch.int64).to(dev)
            parent, tok, score = sel[0], sel[1], k_scores.index_select(0, sel[2])
            host_scores = [host[0][pos] for pos in kept]
        S, Cc = k.cell(S, Cc, g, parent, tok)
        alpha_prev = alpha.index_select(0, parent).contiguous()
        seqs = [seqs[h] + [t] for h, t in zip(parents, toks)]
        trace.append(([(list(q), float(v)) for q, v in zip(seqs, host_scores)], list(done)))
    slots = [0.0] * W
    for i, (_, sc) in enumerate(done[:W]):
        slots[i] = sc
    best_slot = max(range(W), key=lambda i: (slots[i], -i))      # argmax, first on ties
    return (list(done[best_slot][0]) if best_slot < len(done) else []), trace
