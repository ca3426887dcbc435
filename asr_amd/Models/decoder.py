"""Location-aware attention decoder on the HIP kernels -- mirror of
Models/decoder.py:14-61 (+ greedy subset of :63-130) and Models/attention.py:6-40.

Same parameter names as the reference (``att.L_se.weight`` ... ``L_gs.bias``),
same ``forward(hbatch, lengths, targets) -> youtput (B,L,V)``.  The L-step loop
and its BPTT run inside libasr_amd.so (asr_attdec_fwd / asr_attdec_bwd); the
matrix products that are not on the recurrent chain are hoisted into MFMA GEMMs.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib, ops, streams
from .._lib import AttDecBwdArgs, AttDecFwdArgs, check
from ..utils.utils import lengths_to_device
from .attention import Attention

NC, KW = 10, 100


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fill_fwd_args(a, dims, t):
    a.B, a.T, a.E, a.Hd, a.L, a.V = dims
    for k, v in t.items():
        setattr(a, k, v.data_ptr())


def batch_parts(lib, B, T, E, Hd, L):
    """[(b0, b1), ...]: the batch as ONE part when the persistent loop kernels take it (asr_attdec_plan), else the
    fewest equal parts that each do -- B x ceil(T/64) workgroups must each own a compute unit, which BASELINE config 4
    (B = 32, T = 800: 416 workgroups) misses; utterances are independent inside the loop (Models/decoder.py:52-60), so
    two launches of 16 utterances give the same numbers as one of 32.  One part as well when nothing fits: the
    library then runs its four-launches-per-step loop over the whole batch."""
    if lib.asr_attdec_plan(B, T, E, Hd, L) == 3:
        return [(0, B)]
    for n in range(2, min(B, 8) + 1):
        bc = -(-B // n)
        if lib.asr_attdec_plan(bc, T, E, Hd, L) == 3 and lib.asr_attdec_plan(B - bc * (n - 1), T, E, Hd, L) == 3:
            return [(i * bc, min(B, (i + 1) * bc)) for i in range(n) if i * bc < B]
    return [(0, B)]


class _AttDecFunction(torch.autograd.Function):
    # parameter order (state_dict order of the reference's Decoder)
    NAMES = ("W_se", "W_he", "b_he", "w_ee", "W_fe", "conv_w",
             "W_sy", "W_gy", "b_gy", "W_yy", "b_yy", "Emb", "W_ss", "W_gs", "b_gs")
    TRANSPOSED = ("W_se", "W_ss", "W_sy", "W_gs", "W_gy")     # read transposed by asr_attdec_bwd
    @staticmethod
    def forward(ctx, hbatch, lengths_dev, targets, *params):
        lib = _lib.load()
        P = dict(zip(_AttDecFunction.NAMES, params))
        hbatch = hbatch.contiguous()
        targets = targets.contiguous()
        B, T, E = hbatch.shape
        L = targets.shape[1]
        Hd = P["W_sy"].shape[0]
        V = P["W_yy"].shape[0]
        Y = torch.empty(B, L, V, device=hbatch.device, dtype=torch.float32)
        # The backward loop reads five weights transposed.  They do not change between this forward and its backward:
        # when the side stream is on, transpose them there NOW, under the forward loop, instead of in front of the
        # backward loop on the critical path (5 launches, 25 us at the config-2 size).
        ctx.WT = None
        if streams.ENABLED and any(ctx.needs_input_grad):       # (grad mode is always off inside Function.forward)
            dev = hbatch.device
            side, main = streams.side_stream(dev), torch.cuda.current_stream(dev)
            side.wait_stream(main)              # the parameters' last writer (the optimizer) ran on this stream
            with torch.cuda.stream(side):
                wt = tuple(ops.transpose(P[n]) for n in _AttDecFunction.TRANSPOSED)
            ev = torch.cuda.Event()
            ev.record(side)
            ctx.WT = (wt, ev)
        ctx.parts = []
        for b0, b1 in batch_parts(lib, B, T, E, Hd, L):
            t, U = _AttDecFunction._forward_part(lib, P, hbatch[b0:b1], lengths_dev[b0:b1], targets[b0:b1], Y[b0:b1])
            ctx.parts.append((b0, b1, t, U))
        ctx.dims = (B, T, E, Hd, L, V)
        ctx.P = P
        return Y

    @staticmethod
    def _forward_part(lib, P, hbatch, lengths_dev, targets, Y):
        """The loop over one part of the batch (all of it, normally): hbatch, lengths, targets, Y are contiguous
        row ranges of the caller's tensors."""
        B, T, E = hbatch.shape
        L = targets.shape[1]
        Hd = P["W_sy"].shape[0]
        A = 2 * Hd
        V = P["W_yy"].shape[0]
        dev = hbatch.device
        f32 = dict(device=dev, dtype=torch.float32)
        # hoisted: PH = L_he(hbatch)  (Models/attention.py:30 redoes this every step)
        PH = torch.empty(B, T, A, **f32)
        ops.linear_nt(hbatch.view(B * T, E), P["W_he"], PH.view(B * T, A), bias1=P["b_he"])
        S = torch.empty(L + 1, B, Hd, **f32)
        Cst = torch.empty(L + 1, B, Hd, **f32)
        alpha = torch.empty(L, B, T, **f32)
        sproj = torch.empty(L, B, A, **f32)
        g = torch.empty(L, B, E, **f32)
        gates = torch.empty(L, B, 4 * Hd, **f32)
        Fc = torch.empty(L, B, T, NC, **f32)
        ws = torch.empty(int(lib.asr_attdec_ws_floats(B, T, E, Hd, L)), **f32)
        a = AttDecFwdArgs()
        tensors = dict(hbatch=hbatch, PH=PH, lengths=lengths_dev, targets=targets,
                       W_se=P["W_se"], w_ee=P["w_ee"], W_fe=P["W_fe"], conv_w=P["conv_w"],
                       Emb=P["Emb"], W_ss=P["W_ss"], W_gs=P["W_gs"], b_gs=P["b_gs"],
                       S=S, C=Cst, alpha=alpha, sproj=sproj, g=g, gates=gates, Fc=Fc, ws=ws)
        _fill_fwd_args(a, (B, T, E, Hd, L, V), tensors)
        check(lib.asr_attdec_fwd(C.byref(a), _stream()))
        # a timed-out in-launch wait of the persistent loop kernel turns the first context into NaN (-> youtput,
        # loss): the training loop's own NaN check ends the job, as in the reference (train.py:66-68)
        check(lib.asr_poison_if_flag(ws.data_ptr() + 4 * int(lib.asr_attdec_err_offset(B, T, E, Hd, L)),
                                     g.data_ptr(), _stream()))
        # hoisted: U = tanh(L_gy g + L_sy s), youtput = L_yy U   (Models/decoder.py:55)
        U = torch.empty(L, B, Hd, **f32)
        ops.linear_nt(g.view(L * B, E), P["W_gy"], U.view(L * B, Hd), bias1=P["b_gy"])
        ops.linear_nt(S[:L].view(L * B, Hd), P["W_sy"], U.view(L * B, Hd), accumulate=True, act=1)
        ops.sgemm(U, P["W_yy"], Y, B * L, V, Hd, a_two=(L, B * Hd, Hd), b_stride=Hd, c_stride=V,
                  bias1=P["b_yy"])
        return tensors, U

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        if ctx.parts is None:
            raise RuntimeError("asr_amd.Decoder: backward through the decoder a second time -- its saved "
                               "activations are released by the first pass (retain_graph is not supported)")
        B, T, E, Hd, L, V = ctx.dims
        dY = dY.contiguous()
        need_dh = ctx.needs_input_grad[0]
        dh = torch.empty(B, T, E, device=dY.device, dtype=torch.float32) if need_dh else None
        G = None
        if ctx.WT is not None:                  # transposed on the side stream during the forward
            WT, ev = ctx.WT
            main = torch.cuda.current_stream(dY.device)
            main.wait_event(ev)
            for w in WT:
                w.record_stream(main)
        else:
            WT = tuple(ops.transpose(ctx.P[n]) for n in _AttDecFunction.TRANSPOSED)
        for b0, b1, t, U in ctx.parts:
            Gp = _AttDecFunction._backward_part(lib, ctx.P, t, U, (b1 - b0, T, E, Hd, L, V), dY[b0:b1],
                                                dh[b0:b1] if need_dh else None, WT)
            if G is None:
                G = Gp
                continue
            # a further part of the batch: its weight gradients add to the first part's, on the stream that produced
            # both (every weight gradient of this node comes out of an on_side block)
            names = _AttDecFunction.NAMES
            with streams.on_side(dY.device, *[Gp[n] for n in names], *[G[n] for n in names], params=tuple(ctx.P.values())):
                torch._foreach_add_([G[n] for n in names], [Gp[n] for n in names])
        ctx.parts = ctx.WT = None
        return (dh, None, None) + tuple(G[n] for n in _AttDecFunction.NAMES)

    @staticmethod
    def _backward_part(lib, P, t, U, dims, dY, dh, WT):
        """Gradients of one part of the batch: fills dh (when not None), returns the weight gradients by name."""
        B, T, E, Hd, L, V = dims
        A = 2 * Hd
        dev = dY.device
        f32 = dict(device=dev, dtype=torch.float32)
        S, g, alpha, hbatch = t["S"], t["g"], t["alpha"], t["hbatch"]
        u_two = (L, B * Hd, Hd)          # row (b*L + l) of youtput  <->  row (l*B + b) of U
        G = {}
        dU = torch.empty(L, B, Hd, **f32)
        ops.sgemm(dY, P["W_yy"], dU, B * L, Hd, V, a_stride=V, b_mode=1, b_stride=Hd, c_two=u_two)
        dyp = ops.mul_dtanh(dU, U)
        # the loop
        W_seT, W_ssT, W_syT, W_gsT, W_gyT = WT
        drec = torch.empty(L, B, 4 * Hd, **f32)
        dsp = torch.empty(L, B, A, **f32)
        dg = torch.empty(L, B, E, **f32)
        de = torch.empty(L, B, T, **f32)
        dF = torch.empty(L, B, T, NC, **f32)
        dPH = torch.empty(B, T, A, **f32)
        dw_ee = torch.empty(1, A, **f32)
        dW_fe = torch.empty(A, NC, **f32)
        dconv = torch.empty(NC, 1, KW, **f32)
        ws = torch.empty(int(lib.asr_attdec_bwd_ws_floats(B, T, E, Hd, L)), **f32)
        a = AttDecBwdArgs()
        _fill_fwd_args(a.f, dims, t)
        for k, v in dict(W_seT=W_seT, W_ssT=W_ssT, W_syT=W_syT, W_gsT=W_gsT, W_gyT=W_gyT, dyp=dyp, drec=drec,
                         dsp=dsp, dg=dg, de=de, dF=dF, dPH=dPH, dw_ee=dw_ee, dW_fe=dW_fe, dconv_w=dconv,
                         ws=ws).items():
            setattr(a, k, v.data_ptr())
        a.defer_tail = 1        # dw_ee, dW_fe, dconv: asr_attdec_bwd_tail in the side block below
        check(lib.asr_attdec_bwd(C.byref(a), _stream()))
        # a timed-out in-launch wait of the persistent loop kernel turns dPH (-> every encoder gradient) into NaN
        check(lib.asr_poison_if_flag(ws.data_ptr() + 4 * int(lib.asr_attdec_bwd_err_offset(B, T, E, Hd, L)),
                                     dPH.data_ptr(), _stream()))
        dPH2 = dPH.view(B * T, A)
        h2 = hbatch.view(B * T, E)
        # critical path first: d hbatch = dPH W_he + sum_s alpha_s (x) dg_s  (per utterance:
        # (T x L) @ (L x E)) feeds the encoder backward
        if dh is not None:
            ops.matmul_nn(dPH2, P["W_he"], dh.view(B * T, E))
            ops.sgemm(alpha, dg, dh, T, E, L, a_mode=1, a_stride=B * T, b_mode=1, b_stride=B * E, c_stride=E,
                      accumulate=True, nbatch=B, batch_a=T, batch_b=E, batch_c=T * E)
        # hoisted weight gradients: (L*B)-row TN GEMMs over the saved per-step tensors; nothing
        # downstream reads them -> side stream, overlapping the encoder's backward recurrence
        S_in = S[:L].view(L * B, Hd)
        g2, drec2, dyp2, dsp2 = g.view(L * B, E), drec.view(L * B, 4 * Hd), dyp.view(L * B, Hd), dsp.view(L * B, A)

        def tn(a2, b2):
            out = torch.empty(a2.shape[1], b2.shape[1], **f32)
            ops.matmul_tn(a2, b2, out)
            return out

        idx = t["targets"].t().contiguous().view(-1)
        with streams.on_side(dev, S, g, drec, dyp, dsp, dPH, hbatch, idx, dY, U, ws, dF, alpha, de, params=tuple(P.values())):
            # L_yy
            G["W_yy"] = torch.empty(V, Hd, **f32)
            sk = ops.pick_splitk(V, Hd, B * L)
            if sk > 1:
                G["W_yy"].zero_()
            ops.sgemm(dY, U, G["W_yy"], V, Hd, B * L, a_mode=1, a_stride=V, b_mode=1, b_two=u_two, c_stride=Hd,
                      splitk=sk)
            G["b_yy"] = torch.empty(V, **f32)
            ops.colsum(dY.view(B * L, V), G["b_yy"])
            # the attention's own small weights (reads ws, dF, alpha)
            check(lib.asr_attdec_bwd_tail(C.byref(a), _stream()))
            G["W_gs"] = tn(drec2, g2)
            G["W_ss"] = tn(drec2, S_in)
            G["b_gs"] = torch.empty(4 * Hd, **f32)
            ops.colsum(drec2, G["b_gs"])
            G["W_gy"] = tn(dyp2, g2)
            G["W_sy"] = tn(dyp2, S_in)
            G["b_gy"] = torch.empty(Hd, **f32)
            ops.colsum(dyp2, G["b_gy"])
            G["W_se"] = tn(dsp2, S_in)
            G["Emb"] = torch.empty(V, 4 * Hd, **f32)
            ops.scatter_rows(drec2, idx, G["Emb"])
            G["W_he"] = tn(dPH2, h2)
            G["b_he"] = torch.empty(A, **f32)
            ops.colsum(dPH2, G["b_he"])
        G["w_ee"], G["W_fe"], G["conv_w"] = dw_ee, dW_fe, dconv
        assert set(G) == set(_AttDecFunction.NAMES)
        return G


class Decoder(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        hd = hp.num_hidden_nodes_decoder
        he = hp.num_hidden_nodes_encoder * (1 if getattr(hp, "encoder_type", None) == "Conformer" else 2)   # attention.py:13-16, decoder.py:19-22
        self.num_decoder_hidden_nodes = hd
        self.num_encoder_hidden_nodes = hp.num_hidden_nodes_encoder
        self.num_classes = hp.num_classes
        self.att = Attention(hp)
        # prediction (Models/decoder.py:25-28)
        self.L_sy = nn.Linear(hd, hd, bias=False)
        self.L_gy = nn.Linear(he, hd)
        self.L_yy = nn.Linear(hd, hp.num_classes)
        # recurrence (Models/decoder.py:30-33)
        self.L_ys = nn.Embedding(hp.num_classes, hd * 4)
        self.L_ss = nn.Linear(hd, hd * 4, bias=False)
        self.L_gs = nn.Linear(he, hd * 4)

    def _params(self):
        at = self.att
        return (at.L_se.weight, at.L_he.weight, at.L_he.bias, at.L_ee.weight, at.L_fe.weight,
                at.F_conv1d.weight, self.L_sy.weight, self.L_gy.weight, self.L_gy.bias, self.L_yy.weight,
                self.L_yy.bias, self.L_ys.weight, self.L_ss.weight, self.L_gs.weight, self.L_gs.bias)

    def forward(self, hbatch, lengths, targets):
        if not hbatch.is_cuda:
            raise RuntimeError("asr_amd.Decoder: input must be on the GPU; the HIP path has no CPU fallback")
        lengths_dev = lengths_to_device(lengths, hbatch.device)
        return _AttDecFunction.apply(hbatch, lengths_dev, targets, *self._params())

    @torch.no_grad()
    def decode(self, hbatch, lengths, model_lm=None):
        """Models/decoder.py:63-130 (batch 1): greedy for ``hp.beam_width == 1``, beam search otherwise.

        Runs the same per-step kernels as training.  Returns the best hypothesis once it ends in
        ``hp.eos_id`` within ``hp.max_decoder_seq_len`` steps, else ``[]`` -- exactly what the
        reference returns.  ``model_lm`` (shallow fusion, :93-96): a callable ``tokens (n, len) int64 on the device
        -> scores (n, len, V)``, weighted by ``hp.lm_weight``; log_softmax scoring only (beam.py::_fuse_lm).
        """
        width = int(getattr(self.hp, "beam_width", 1))
        if width > 1 or model_lm is not None:
            from .beam import beam_decode
            return beam_decode(self, hbatch, lengths, width, self.hp.max_decoder_seq_len, self.hp.eos_id,
                               getattr(self.hp, "score_func", "log_softmax"), model_lm,
                               self.hp.lm_weight if model_lm is not None else 0.0)[0]
        return self.decode_raw(hbatch, lengths)[0]

    @torch.no_grad()
    def decode_v2(self, hbatch, lengths, model_lm=None):
        """Models/decoder.py:132-267 (batch 1): the second beam search -- finished
        hypotheses are collected in ``hp.beam_width`` result slots and the best-scoring slot is
        returned (see ``beam.beam_decode_v2`` for the rules, including when [] comes back).  ``model_lm``: shallow
        fusion (:180-183), as in ``decode``.  The
        reference's trailing matplotlib / sentencepiece imports (:261-264) have no effect on the
        result and are not mirrored."""
        from .beam import beam_decode_v2
        return beam_decode_v2(self, hbatch, lengths, int(self.hp.beam_width), self.hp.max_decoder_seq_len,
                              self.hp.eos_id, getattr(self.hp, "score_func", "log_softmax"), model_lm,
                              self.hp.lm_weight if model_lm is not None else 0.0)[0]

    @torch.no_grad()
    def decode_raw(self, hbatch, lengths):
        assert hbatch.shape[0] == 1, "greedy decode is batch-1, as in the reference"
        from .greedy import greedy_decode
        return greedy_decode(self, hbatch, lengths, self.hp.max_decoder_seq_len, self.hp.eos_id)
