"""Drop-in for the Conformer branch of Models/Transformer_encoder.py (TransformerEncoder, :441-451:
CNN_embedding :386-407 + ConformerEncoder :348-365) on the HIP kernels.

Same constructor arguments, same parameter / buffer names (``cnn_encoder.conv.0.weight`` ...
``encoder.layers.15.ff_2.linear2.bias``) and the same default initialisation order as the
reference, so checkpoints and seeded initialisations interchange.  The torch modules below
(``nn.Linear``, ``nn.Conv1d`` ...) only *hold* parameters; no torch compute module is ever called.
Every block of a layer is one autograd node with a hand-written backward over libasr_amd.so:
matrix products on ``asr_sgemm``, everything else on the kernels of csrc/conformer.hip.
Activations stay (B, T, C) row-major; no CPU fallback.

Not mirrored: the per-layer attention maps the reference collects and drops (``attns_enc``, :361-364).
"""
import ctypes as C
import functools
import math
import os

import torch
import torch.nn as nn

from .. import _lib, ops, streams
from .._lib import check

KSIZE = 31      # Transformer_encoder.py:215
LN_EPS = 1e-5   # nn.LayerNorm / nn.BatchNorm1d defaults
BN_MOMENTUM = 0.1


# Arithmetic of this encoder's matrix products (ops.gemm_precision): the bf16x3 split, like the rest of the model,
# except the FORWARD products of the Conv2d subsampling, which stay exact fp32.  Measured at the config-5 size against
# the all-exact-fp32 step (tools/prec_conformer.py, profiles/r2_conformer_precision.txt): every gradient within 9e-5 of
# its tensor's largest entry.  With the subsampling's forward split too, its two conv weights land at 1.3e-3 / 3.1e-3:
# a pre-activation that round-off moves across zero flips its ReLU mask, a whole term of a ~1e6-term, heavily
# cancelling sum (fp32 against float64 shows the same effect at its own scale, tools/cnn_check.py).
# This file INHERITS the process-wide mode (ASR_GEMM_PRECISION / ops.gemm_precision, default bf16x3): a caller who
# asked for the reference's exact fp32 products gets them here too.  ASR_CONFORMER_GEMM_PRECISION, only when set,
# overrides the mode for this encoder alone (A/B runs).
GEMM_PRECISION = os.environ.get("ASR_CONFORMER_GEMM_PRECISION")


def _exact(fn):
    """Run a forward / backward of this file under GEMM_PRECISION when that override is set."""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        if GEMM_PRECISION is None:
            return fn(*a, **k)
        with ops.gemm_precision(GEMM_PRECISION):
            return fn(*a, **k)
    return wrapped


# The forward products of the two Conv2d of the subsampling -- the ones in front of a ReLU -- keep a setting of their
# own (see the top of the file); its output Linear takes the general mode.
CNN_FWD_GEMM_PRECISION = os.environ.get("ASR_CONFORMER_CNN_FWD_GEMM_PRECISION", "f32")


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _e(*shape, like, dtype=torch.float32):
    return torch.empty(*shape, device=like.device, dtype=dtype)


# ------------------------------------------------------------------ kernel wrappers --
def _ln_fwd(x2, g, b):
    rows, d = x2.shape
    y, mean, rstd = torch.empty_like(x2), _e(rows, like=x2), _e(rows, like=x2)
    check(_lib.load().asr_layernorm_fwd(x2.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), rows, d, LN_EPS, _st()))
    return y, mean, rstd


def _ln_bwd(dy2, x2, g, mean, rstd, res):
    """Returns (res + d LayerNorm / dx, per-block partials of [d gamma | d beta]); ``res`` = the gradient over the
    residual connection.  The partials are summed by _ln_bwd_sums inside the caller's side-stream block: nothing
    downstream reads d gamma / d beta, and the two launches of the sum were 64 times per step on the chain."""
    lib = _lib.load()
    rows, d = x2.shape
    nb = int(lib.asr_layernorm_bwd_blocks(rows))
    part = _e(nb, 2 * d, like=x2)
    dx = torch.empty_like(x2)
    check(lib.asr_layernorm_bwd(dy2.data_ptr(), x2.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                dx.data_ptr(), part.data_ptr(), rows, d, res.data_ptr(), _st()))
    return dx, part


def _ln_bwd_sums(part):
    d = part.shape[1] // 2
    gb = _e(2 * d, like=part)
    ops.colsum(part, gb)
    return gb[:d], gb[d:]


def _residual_drop(x2, o, factor, p, training, ndrop):
    """y = x + factor * drop^ndrop(o) in one pass; returns (y, mask or None, the factor of the gradient)."""
    lib = _lib.load()
    y = torch.empty_like(x2)
    on = training and p > 0.0 and ndrop > 0
    mask = torch.empty(x2.shape, device=x2.device, dtype=torch.uint8) if on else None
    f = factor / (1.0 - p) ** ndrop if on else factor
    check(lib.asr_residual_dropout(x2.data_ptr(), o.data_ptr(), y.data_ptr(), mask.data_ptr() if on else 0, x2.numel(),
                                   C.c_float(f), C.c_float(p if on else 0.0), C.c_uint64(_seed(True, 1.0) if on else 0),
                                   ndrop if on else 0, _st()))
    return y, mask, f


def _scale_mask(dy2, mask, f):
    if mask is None and f == 1.0:
        return dy2
    out = torch.empty_like(dy2)
    check(_lib.load().asr_scale_mask(dy2.data_ptr(), mask.data_ptr() if mask is not None else 0, out.data_ptr(),
                                     dy2.numel(), C.c_float(f), _st()))
    return out


def _linear(x2, W2, b=None, act=0):
    y = _e(x2.shape[0], W2.shape[0], like=x2)
    ops.linear_nt(x2, W2, y, bias1=b, act=act)
    return y


def _zeros_like_all(*ws):
    """One zero fill for several weight gradients (the split-K products add into them): views of one flat buffer."""
    flat = torch.zeros(sum(w.numel() for w in ws), device=ws[0].device, dtype=torch.float32)
    out, o = [], 0
    for w in ws:
        out.append(flat[o:o + w.numel()].view(w.shape))
        o += w.numel()
    return out


def _linear_dx(dy2, W2, dx_out=None, accumulate=False):
    """dx of y = x W^T + b, written (or added) into dx_out when given: the part of a Linear's backward the next
    kernel of the chain waits for."""
    dx = dx_out if dx_out is not None else _e(dy2.shape[0], W2.shape[1], like=dy2)
    ops.matmul_nn(dy2, W2, dx, accumulate=accumulate)
    return dx


def _linear_dw(pairs, bias_pairs):
    """The part nothing downstream reads: dW = dy^T x for every (dy2, x2, W2) of ``pairs`` (split-K products into one
    zero-filled buffer) and the column sums ``bias_pairs`` = [(dy2, width)] in one ops.colsum_multi call.  Called
    inside a streams.on_side block by the blocks' backward functions, so that these products and sums overlap the
    NEXT block's data-gradient chain (layer norm, activations, dropout masks: HBM-bound passes that leave the matrix
    pipe idle).  Returns ([dW], [db])."""
    dWs = _zeros_like_all(*[w for _, _, w in pairs])
    for (dy2, x2, _), dW in zip(pairs, dWs):
        ops.matmul_tn(dy2, x2, dW, zeroed=True)
    dbs = [_e(n, like=dy2) for dy2, n in bias_pairs]
    if dbs:
        ops.colsum_multi([(dy2, db) for (dy2, _), db in zip(bias_pairs, dbs)])
    return dWs, dbs


def _drop(x, p, training, seed):
    if not training or p <= 0.0:
        return x, None
    y, mask = torch.empty_like(x), torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(_lib.load().asr_dropout_fwd(x.data_ptr(), y.data_ptr(), mask.data_ptr(), x.numel(), C.c_float(p),
                                      C.c_uint64(seed), _st()))
    return y, mask


def _drop_bwd(dy, mask, p):
    if mask is None:
        return dy
    dx = torch.empty_like(dy)
    check(_lib.load().asr_dropout_bwd(dy.data_ptr(), mask.data_ptr(), dx.data_ptr(), dy.numel(), C.c_float(p), _st()))
    return dx


def _axpby(x, y, a, b, out=None):
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().asr_axpby(x.data_ptr(), y.data_ptr(), out.data_ptr(), x.numel(), C.c_float(a), C.c_float(b), _st()))
    return out


def _seed(training, p):
    return int(torch.randint(0, 2 ** 62, (1,)).item()) if (training and p > 0.0) else 0


def _req(x):
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("asr_amd Conformer encoder: input must be a float32 tensor on the GPU; "
                           "the HIP path has no CPU fallback")
    return x.contiguous()


# ------------------------------------------------------------ feed-forward block --
class _FFFunction(torch.autograd.Function):
    """y = x + scale * drop^n(linear2(drop(swish(linear1(LN(x))))))  (FeedForwardConformer :191-209
    inside ConformerEncoderLayer :259,265; n = 1 for ff_1, 2 for ff_2 whose output passes dropout_2 too)."""

    @staticmethod
    @_exact
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, scale, p, training, n_out_drop):
        lib = _lib.load()
        x2 = _req(x).view(-1, x.shape[-1])
        on = training and p > 0.0
        h, mean, rstd = _ln_fwd(x2, ln_w, ln_b)
        a = _linear(h, w1, b1)
        s = torch.empty_like(a)
        m1 = torch.empty(a.shape, device=a.device, dtype=torch.uint8) if on else None
        check(lib.asr_swish_fwd(a.data_ptr(), s.data_ptr(), m1.data_ptr() if on else 0, a.numel(),
                                C.c_float(p if on else 0.0), C.c_uint64(_seed(training, p)), _st()))
        o = _linear(s, w2, b2)
        y, m_o, f = _residual_drop(x2, o, scale, p, training, n_out_drop)
        ctx.save_for_backward(x2, ln_w, w1, w2, mean, rstd, h, a, s)
        ctx.aux = (p, m1, m_o, f)
        ctx.wparams = (w1, b1, w2, b2, ln_w, ln_b)
        return y.view_as(x)

    @staticmethod
    @_exact
    def backward(ctx, dy):
        lib = _lib.load()
        x2, ln_w, w1, w2, mean, rstd, h, a, s = ctx.saved_tensors
        p, m1, m_o, f = ctx.aux
        dy2 = dy.contiguous().view_as(x2)
        do = _scale_mask(dy2, m_o, f)
        ds = _linear_dx(do, w2)
        da = torch.empty_like(a)
        check(lib.asr_swish_bwd(ds.data_ptr(), a.data_ptr(), m1.data_ptr() if m1 is not None else 0, da.data_ptr(),
                                a.numel(), C.c_float(p if m1 is not None else 0.0), _st()))
        dh = _linear_dx(da, w1)
        dx, lnpart = _ln_bwd(dh, x2, ln_w, mean, rstd, dy2)
        with streams.on_side(dy2.device, do, s, da, h, lnpart, params=ctx.wparams):
            (dW2, dW1), (db2, db1) = _linear_dw([(do, s, w2), (da, h, w1)], [(do, w2.shape[0]), (da, w1.shape[0])])
            dg, dbeta = _ln_bwd_sums(lnpart)
        return dx.view_as(dy), dg, dbeta, dW1, db1, dW2, db2, None, None, None, None


# ------------------------------------------------- relative-position attention block --
def _head_gemm(A, Bm, Cm, M, N, K, H, nb, *, a, b, c, a_mode=0, b_mode=0, pad_ok=0):
    """One product per (utterance, head): a / b / c = (row stride, per-head offset, per-utterance offset)."""
    ops.sgemm(A, Bm, Cm, M, N, K, a_mode=a_mode, b_mode=b_mode, a_stride=a[0], b_stride=b[0], c_stride=c[0],
              nbatch=nb * H, batch_a=a[1], batch_b=b[1], batch_c=c[1], batch_two=(H, a[2], b[2], c[2]), pad_ok=pad_ok)


def _pad4(T):
    """Row stride of the attention maps: the next multiple of 4, so the products that read them
    column-wise (attn^T d ctx, d ac^T (q + u), d bd^T (q + v)) use 16-byte loads (asr_gemm_desc.pad_ok)."""
    return (T + 3) // 4 * 4


class _RelAttnFunction(torch.autograd.Function):
    """y = x + drop(out(attention(LN(x))))  (ConformerEncoderLayer :260-263 with
    RelativeMultiHeadAttention :76-137)."""

    @staticmethod
    @_exact
    def forward(ctx, x, nw, nb_, wq, bq, wk, bk, wv, bv, wpos, u, v, wo, bo, pe, keep, H, p, training):
        lib = _lib.load()
        B, T, d = x.shape
        dk = d // H
        x2 = _req(x).view(B * T, d)
        seed = _seed(training, p)
        h, mean, rstd = _ln_fwd(x2, nw, nb_)
        q, k, vv = _linear(h, wq, bq), _linear(h, wk, bk), _linear(h, wv, bv)
        pp = _linear(pe, wpos)                                              # (T, d), shared by the batch
        qu, qv = torch.empty_like(q), torch.empty_like(q)
        check(lib.asr_add_two_bias(q.data_ptr(), u.data_ptr(), v.data_ptr(), qu.data_ptr(), qv.data_ptr(), B * T, d, _st()))
        Tp = _pad4(T)
        proj, score, posp = (d, dk, T * d), (Tp, T * Tp, H * T * Tp), (d, dk, 0)
        attn = _e(B, H, T, Tp, like=x)
        bd = _e(B, H, T, Tp, like=x)
        _head_gemm(qu, k, attn, T, T, dk, H, B, a=proj, b=proj, c=score)      # (q + u) k^T          :97
        _head_gemm(qv, pp, bd, T, T, dk, H, B, a=proj, b=posp, c=score)       # (q + v) p^T          :98
        scale = 1.0 / math.sqrt(dk)
        on = training and p > 0.0
        attn_d = torch.empty_like(attn) if on else attn
        m_att = torch.empty(attn.shape, device=x.device, dtype=torch.uint8) if on else None
        check(lib.asr_relattn_softmax_fwd(attn.data_ptr(), bd.data_ptr(), keep.data_ptr(), C.c_float(scale), B, H, T, Tp,
                                          attn_d.data_ptr() if on else 0, m_att.data_ptr() if on else 0,
                                          C.c_float(p if on else 0.0), C.c_uint64(seed), _st()))
        del bd
        cx = _e(B * T, d, like=x)
        # (the maps' padding columns are zero: pad_ok bit 2 lets K = T, not a multiple of 4, take the 16-byte path)
        _head_gemm(attn_d, vv, cx, T, dk, T, H, B, a=score, b=proj, c=proj, b_mode=1, pad_ok=4)   # attn v   :136
        y, m_o, f_o = _residual_drop(x2, _linear(cx, wo, bo), 1.0, p, training, 1)
        ctx.save_for_backward(x2, nw, wq, wk, wv, wpos, wo, pe, keep, mean, rstd, h, qu, qv, k, vv, pp, attn, attn_d, cx)
        ctx.aux = (B, T, d, H, p, scale, m_att, m_o, f_o)
        ctx.wparams = (wq, bq, wk, bk, wv, bv, wpos, u, v, wo, bo, nw, nb_)
        return y.view_as(x)

    @staticmethod
    @_exact
    def backward(ctx, dy):
        lib = _lib.load()
        (x2, nw, wq, wk, wv, wpos, wo, pe, keep, mean, rstd, h, qu, qv, k, vv, pp, attn, attn_d, cx) = ctx.saved_tensors
        B, T, d, H, p, scale, m_att, m_o, f_o = ctx.aux
        dk = d // H
        Tp = _pad4(T)
        proj, score, posp = (d, dk, T * d), (Tp, T * Tp, H * T * Tp), (d, dk, 0)
        dy2 = dy.contiguous().view_as(x2)
        do = _scale_mask(dy2, m_o, f_o)
        dcx = _linear_dx(do, wo)
        dattn = _e(B, H, T, Tp, like=x2)
        _head_gemm(dcx, vv, dattn, T, T, dk, H, B, a=proj, b=proj, c=score)                       # d attn = d ctx v^T
        dvv = torch.empty_like(vv)
        _head_gemm(attn_d, dcx, dvv, T, dk, T, H, B, a=score, b=proj, c=proj, a_mode=1, b_mode=1, pad_ok=1)  # d v = attn^T d ctx
        dbd = torch.empty_like(dattn)
        check(lib.asr_relattn_softmax_bwd(attn.data_ptr(), dattn.data_ptr(), dbd.data_ptr(), keep.data_ptr(),
                                          C.c_float(scale), B, H, T, Tp, m_att.data_ptr() if m_att is not None else 0,
                                          C.c_float(p if m_att is not None else 0.0), _st()))
        dqu, dkk, dqv = torch.empty_like(qu), torch.empty_like(k), torch.empty_like(qv)
        _head_gemm(dattn, k, dqu, T, dk, T, H, B, a=score, b=proj, c=proj, b_mode=1, pad_ok=4)     # d ac k
        _head_gemm(dattn, qu, dkk, T, dk, T, H, B, a=score, b=proj, c=proj, a_mode=1, b_mode=1, pad_ok=1)    # d ac^T (q + u)
        _head_gemm(dbd, pp, dqv, T, dk, T, H, B, a=score, b=posp, c=proj, b_mode=1, pad_ok=4)      # d bd p
        dq = _axpby(dqu, dqv, 1.0, 1.0)
        dh = _linear_dx(dq, wq)
        _linear_dx(dkk, wk, dx_out=dh, accumulate=True)
        _linear_dx(dvv, wv, dx_out=dh, accumulate=True)
        dx, lnpart = _ln_bwd(dh, x2, nw, mean, rstd, dy2)
        with streams.on_side(dy2.device, do, cx, dq, dkk, dvv, h, dbd, qv, pe, dqu, dqv, lnpart, params=ctx.wparams):
            dg, dbeta = _ln_bwd_sums(lnpart)
            # d p[h] = sum_b d bd[b,h]^T (q + v)[b,:,h,:]: the reduction runs over (b, i) through a two-level row map
            # (only heads x ceil(T/64) tiles: split the long reduction over workgroups, 464 -> 52 us at config 5)
            sk = max(1, min(16, (B * T) // 512))
            dpp = torch.zeros_like(pp) if sk > 1 else torch.empty_like(pp)
            ops.sgemm(dbd, qv, dpp, T, dk, B * T, a_mode=1, b_mode=1, a_two=(T, Tp, H * T * Tp), b_stride=d, c_stride=d,
                      nbatch=H, batch_a=T * Tp, batch_b=dk, batch_c=dk, pad_ok=1, splitk=sk)
            (dWo, dWq, dWk, dWv, dWpos), (dbo, dbq, dbk, dbv, du, dv) = _linear_dw(
                [(do, cx, wo), (dq, h, wq), (dkk, h, wk), (dvv, h, wv), (dpp, pe, wpos)],
                [(do, d), (dq, d), (dkk, d), (dvv, d), (dqu, d), (dqv, d)])    # six column sums, two launches
        return (dx.view_as(dy), dg, dbeta, dWq, dbq, dWk, dbk, dWv, dbv, dWpos, du.view(H, dk), dv.view(H, dk), dWo, dbo,
                None, None, None, None, None)


# ------------------------------------------------------------ convolution module --
class _ConvModuleFunction(torch.autograd.Function):
    """y = x + drop(pw2(swish(BN(conv_out(depthwise(GLU(pw1(LN(x)))))))))  (ConvolutionModule :228-240 inside
    ConformerEncoderLayer :264).  BatchNorm statistics cover all B*T frames, as in the reference."""

    @staticmethod
    @_exact
    def forward(ctx, x, lnw, lnb, pw1w, pw1b, dww, dwb, cow, cob, bnw, bnb, pw2w, pw2b, run_mean, run_var, p, training):
        lib = _lib.load()
        B, T, d = x.shape
        rows = B * T
        x2 = _req(x).view(rows, d)
        W1, Wd, Wc, W2 = pw1w.view(2 * d, d), dww.view(d, KSIZE), cow.view(d, d), pw2w.view(d, d)
        h, mean, rstd = _ln_fwd(x2, lnw, lnb)
        a = _linear(h, W1, pw1b)
        g = _e(rows, d, like=x)
        check(lib.asr_glu_fwd(a.data_ptr(), g.data_ptr(), rows, d, _st()))
        c1 = torch.empty_like(g)
        check(lib.asr_dwconv(g.data_ptr(), Wd.data_ptr(), dwb.data_ptr(), c1.data_ptr(), B, T, d, KSIZE, 0, _st()))
        c2 = _linear(c1, Wc, cob)
        if training:
            msum, vsum = _e(d, like=x), _e(d, like=x)
            ops.colsum(c2, msum)
            part = _e(int(lib.asr_bn_blocks(rows)), d, like=x)
            check(lib.asr_bn_var_part(c2.data_ptr(), msum.data_ptr(), C.c_float(1.0 / rows), part.data_ptr(), rows, d, _st()))
            ops.colsum(part, vsum)
            ms = vs = 1.0 / rows
            # running statistics: momentum 0.1, unbiased variance (nn.BatchNorm1d defaults, :223)
            _axpby(run_mean, msum, 1.0 - BN_MOMENTUM, BN_MOMENTUM / rows, out=run_mean)
            _axpby(run_var, vsum, 1.0 - BN_MOMENTUM, BN_MOMENTUM / max(rows - 1, 1), out=run_var)
        else:
            msum, vsum, ms, vs = run_mean.clone(), run_var.clone(), 1.0, 1.0
        s = torch.empty_like(c2)
        check(lib.asr_bn_swish_fwd(c2.data_ptr(), msum.data_ptr(), vsum.data_ptr(), C.c_float(ms), C.c_float(vs),
                                   bnw.data_ptr(), bnb.data_ptr(), s.data_ptr(), rows, d, LN_EPS, _st()))
        y, m_o, f_o = _residual_drop(x2, _linear(s, W2, pw2b), 1.0, p, training, 1)
        ctx.save_for_backward(x2, lnw, pw1w, dww, cow, bnw, bnb, pw2w, mean, rstd, h, a, g, c1, c2, msum, vsum, s)
        ctx.aux = (B, T, d, p, training, ms, vs, m_o, f_o)
        ctx.wparams = (pw1w, pw1b, dww, dwb, cow, cob, pw2w, pw2b, lnw, lnb)
        return y.view_as(x)

    @staticmethod
    @_exact
    def backward(ctx, dy):
        lib = _lib.load()
        (x2, lnw, pw1w, dww, cow, bnw, bnb, pw2w, mean, rstd, h, a, g, c1, c2, msum, vsum, s) = ctx.saved_tensors
        B, T, d, p, training, ms, vs, m_o, f_o = ctx.aux
        rows = B * T
        W1, Wd, Wc, W2 = pw1w.view(2 * d, d), dww.view(d, KSIZE), cow.view(d, d), pw2w.view(d, d)
        dy2 = dy.contiguous().view_as(x2)
        do = _scale_mask(dy2, m_o, f_o)
        ds = _linear_dx(do, W2)
        part = _e(int(lib.asr_bn_blocks(rows)), 2 * d, like=x2)
        check(lib.asr_bn_swish_bwd_part(ds.data_ptr(), c2.data_ptr(), msum.data_ptr(), vsum.data_ptr(), C.c_float(ms),
                                        C.c_float(vs), bnw.data_ptr(), bnb.data_ptr(), part.data_ptr(), rows, d, LN_EPS, _st()))
        sums = _e(2 * d, like=x2)
        ops.colsum(part, sums)                                              # [d beta, d gamma]
        dc2 = torch.empty_like(c2)
        check(lib.asr_bn_swish_bwd(ds.data_ptr(), c2.data_ptr(), msum.data_ptr(), vsum.data_ptr(), C.c_float(ms),
                                   C.c_float(vs), bnw.data_ptr(), bnb.data_ptr(), sums.data_ptr() if training else 0,
                                   dc2.data_ptr(), rows, d, LN_EPS, _st()))
        dc1 = _linear_dx(dc2, Wc)
        dg = torch.empty_like(g)
        check(lib.asr_dwconv(dc1.data_ptr(), Wd.data_ptr(), 0, dg.data_ptr(), B, T, d, KSIZE, 1, _st()))
        da = torch.empty_like(a)
        check(lib.asr_glu_bwd(dg.data_ptr(), a.data_ptr(), da.data_ptr(), rows, d, _st()))
        dh = _linear_dx(da, W1)
        dx, lnpart = _ln_bwd(dh, x2, lnw, mean, rstd, dy2)
        with streams.on_side(dy2.device, do, s, dc2, c1, dc1, g, da, h, lnpart, params=ctx.wparams):
            dgam, dbeta = _ln_bwd_sums(lnpart)
            wpart = _e(int(lib.asr_dwconv_bwd_w_chunks(B, T)), KSIZE * d, like=x2)
            check(lib.asr_dwconv_bwd_w(dc1.data_ptr(), g.data_ptr(), wpart.data_ptr(), B, T, d, KSIZE, _st()))
            dwkc = _e(KSIZE, d, like=x2)
            ops.colsum(wpart, dwkc.view(-1))
            dWd = ops.transpose(dwkc)                                       # (K, d) -> (d, K)
            (dW2, dWc, dW1), (db2, dbc, ddb, db1) = _linear_dw(
                [(do, s, W2), (dc2, c1, Wc), (da, h, W1)], [(do, d), (dc2, d), (dc1, d), (da, 2 * d)])
        return (dx.view_as(dy), dgam, dbeta, dW1.view_as(pw1w), db1, dWd.view_as(dww), ddb, dWc.view_as(cow), dbc,
                sums[d:], sums[:d], dW2.view_as(pw2w), db2, None, None, None, None)


# ------------------------------------------------------------ Conv2d subsampling --
class _CnnEmbeddingFunction(torch.autograd.Function):
    """CNN_embedding.forward (:400-404): two 3x3 stride-2 Conv2d + ReLU as patch gathers + asr_sgemm (relu
    epilogue), then the bias-free Linear over (channel, frequency).  Channels-last maps, so the Linear's
    weight is used as its (odim, F2, C) permutation.  No gradient w.r.t. the features (they are data)."""

    @staticmethod
    @_exact
    def forward(ctx, src, w1, b1, w2, b2, wout):
        lib = _lib.load()
        B, T, F = src.shape
        Cc = w1.shape[0]
        src = _req(src)
        T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        # the 9 taps of the first convolution in rows of 12 (three zero columns): 16-byte rows for asr_sgemm
        col1 = _e(B * T1 * F1, 12, like=src)
        check(lib.asr_im2col_3x3s2(src.data_ptr(), col1.data_ptr(), B, T, F, 1, 12, _st()))
        w1p = torch.zeros(Cc, 12, device=src.device, dtype=torch.float32)
        w1p[:, :9].copy_(w1.view(Cc, 9))
        with ops.gemm_precision(CNN_FWD_GEMM_PRECISION):                    # the two products in front of a ReLU
            y1 = _linear(col1, w1p, b1, act=2)                              # (B,T1,F1,C)
        col2 = _e(B * T2 * F2, 9 * Cc, like=src)
        check(lib.asr_im2col_3x3s2(y1.data_ptr(), col2.data_ptr(), B, T1, F1, Cc, 9 * Cc, _st()))
        w2p = ops.transpose(w2.view(Cc, Cc, 9)).view(Cc, 9 * Cc)            # (Cout, ki, kj, Cin)
        with ops.gemm_precision(CNN_FWD_GEMM_PRECISION):
            y2 = _linear(col2, w2p, b2, act=2)                              # (B,T2,F2,C)
        woutp = ops.transpose(wout.view(-1, Cc, F2)).view(-1, F2 * Cc)      # (odim, F2, C)
        # the output Linear has no ReLU behind it (no mask for round-off to flip, see CNN_FWD_GEMM_PRECISION): it takes
        # the encoder's general arithmetic like every other Linear of the model
        out = _linear(y2.view(B * T2, F2 * Cc), woutp)
        ctx.save_for_backward(col1, y1, col2, y2, w2p, woutp)
        ctx.dims = (B, T1, F1, T2, F2, Cc)
        ctx.wparams = (w1, b1, w2, b2, wout)
        return out.view(B, T2, -1)

    @staticmethod
    @_exact
    def backward(ctx, dout):
        lib = _lib.load()
        col1, y1, col2, y2, w2p, woutp = ctx.saved_tensors
        B, T1, F1, T2, F2, Cc = ctx.dims
        do = dout.contiguous().view(B * T2, -1)
        y2f = y2.view(B * T2, F2 * Cc)
        dy2 = _linear_dx(do, woutp)
        dpre2 = torch.empty_like(y2)
        check(lib.asr_relu_bwd(dy2.data_ptr(), y2.data_ptr(), dpre2.data_ptr(), y2.numel(), _st()))
        # the two large weight gradients run beside the data gradient of the second convolution and its col2im
        with streams.on_side(do.device, do, y2, dpre2, col2, params=ctx.wparams):
            (dwoutp, dw2p), (db2,) = _linear_dw([(do, y2f, woutp), (dpre2, col2, w2p)], [(dpre2, Cc)])
            dwout = ops.transpose(dwoutp.view(-1, F2, Cc)).view(-1, Cc * F2)
            dw2 = ops.transpose(dw2p.view(Cc, 9, Cc)).view(Cc, Cc, 3, 3)
        dcol2 = _linear_dx(dpre2, w2p)
        if Cc % 4 == 0 and 256 % (Cc // 4) == 0:
            # col2im, the ReLU gradient and the first convolution's weight / bias gradients in ONE pass: d pre1 (957 MB at
            # config 5) is never written -- its only readers were dW1 = d pre1^T col1 and the bias column sum
            part = _e(int(lib.asr_col2im_dw_blocks(B, T1, F1, Cc)), Cc * 10, like=y1)
            check(lib.asr_col2im_3x3s2_dw(dcol2.data_ptr(), y1.data_ptr(), col1.data_ptr(), 12, part.data_ptr(), B, T1, F1,
                                          Cc, _st()))
            wb = _e(Cc * 10, like=y1)
            ops.colsum(part, wb)
            wb = wb.view(Cc, 10)
            dw1p, db1 = wb, wb[:, 9].contiguous()
        else:
            dpre1 = torch.empty_like(y1)                                    # col2im and the ReLU gradient in one pass
            check(lib.asr_col2im_3x3s2(dcol2.data_ptr(), y1.data_ptr(), dpre1.data_ptr(), B, T1, F1, Cc, _st()))
            dw1p = _e(Cc, 12, like=y1)
            ops.matmul_tn(dpre1.view(-1, Cc), col1, dw1p)
            db1 = _e(Cc, like=y1)
            ops.colsum(dpre1.view(-1, Cc), db1)
        return None, dw1p[:, :9].reshape(Cc, 1, 3, 3), db1, dw2, db2, dwout


# ------------------------------------------------------------------- module tree --
class Swish(nn.Module):
    """Parameter-free place holder (Transformer_encoder.py:175-177); the kernels fuse it."""


class DepthwiseConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, padding):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, groups=in_channels)
        self.conv_out = nn.Conv1d(out_channels, out_channels, 1)


class FeedForwardConformer(nn.Module):
    def __init__(self, d_model, d_ff=2048, dropout=0.1):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.p = dropout

    def run(self, x, scale, n_out_drop):
        return _FFFunction.apply(x, self.layer_norm.weight, self.layer_norm.bias, self.linear1.weight, self.linear1.bias,
                                 self.linear2.weight, self.linear2.bias, scale, self.p, self.training, n_out_drop)


class RelativeMultiHeadAttention(nn.Module):
    def __init__(self, heads, d_model, dropout=0.1):
        super().__init__()
        self.d_model, self.d_k, self.h = d_model, d_model // heads, heads
        self.q_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.linear_pos = nn.Linear(d_model, d_model, bias=False)
        self.pos_bias_u = nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.Tensor(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)
        self.out = nn.Linear(d_model, d_model)
        self.p = dropout


class ConvolutionModule(nn.Module):
    def __init__(self, d_model, dropout=0.1):
        super().__init__()
        pad = KSIZE // 2
        self.layer_norm = nn.LayerNorm(d_model)
        self.pointwise_conv1 = nn.Conv1d(d_model, d_model * 2, 1)
        self.depth_conv1 = DepthwiseConv(d_model, d_model, KSIZE, (pad, pad - (KSIZE + 1) % 2))
        self.batch_norm = nn.BatchNorm1d(d_model)
        self.pointwise_conv2 = nn.Conv1d(d_model, d_model, 1)
        self.p = dropout

    def run(self, x):
        bn, dc = self.batch_norm, self.depth_conv1
        y = _ConvModuleFunction.apply(
            x, self.layer_norm.weight, self.layer_norm.bias, self.pointwise_conv1.weight, self.pointwise_conv1.bias,
            dc.conv.weight, dc.conv.bias, dc.conv_out.weight, dc.conv_out.bias, bn.weight, bn.bias,
            self.pointwise_conv2.weight, self.pointwise_conv2.bias, bn.running_mean, bn.running_var, self.p, self.training)
        if self.training:
            bn.num_batches_tracked += 1
        return y


class ConformerEncoderLayer(nn.Module):
    def __init__(self, d_model, heads, dropout=0.1):
        super().__init__()
        self.ff_1 = FeedForwardConformer(d_model, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.attn = RelativeMultiHeadAttention(heads, d_model, dropout=dropout)
        self.conv_module = ConvolutionModule(d_model, dropout=dropout)
        self.ff_2 = FeedForwardConformer(d_model, dropout=dropout)
        self.p = dropout

    def forward(self, x, pe, keep):
        at = self.attn
        x = self.ff_1.run(x, 0.5, 1)                                        # :259
        x = _RelAttnFunction.apply(x, self.norm.weight, self.norm.bias, at.q_linear.weight, at.q_linear.bias,
                                   at.k_linear.weight, at.k_linear.bias, at.v_linear.weight, at.v_linear.bias,
                                   at.linear_pos.weight, at.pos_bias_u, at.pos_bias_v, at.out.weight, at.out.bias,
                                   pe, keep, at.h, self.p, self.training)   # :260-263
        x = self.conv_module.run(x)                                         # :264
        return self.ff_2.run(x, 1.0, 2)                                     # :265


class RelativePositionalEncoder(nn.Module):
    """Transformer_encoder.py:302-323: the sin/cos table for positions 0..T-1, built on the CPU exactly as
    the reference builds it (a plain attribute there too, so it is not in the state_dict) and cached per
    device; xscale = 1; dropout on both the input and the table."""

    def __init__(self, d_model, max_seq_len=5000, dropout=0.1):
        super().__init__()
        self.d_model, self.p = d_model, dropout
        pe = torch.zeros(max_seq_len, d_model)
        position = torch.arange(0, max_seq_len, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.pe = pe.unsqueeze(0)
        self._dev = {}

    def table(self, T, device):
        if device not in self._dev:
            self._dev[device] = self.pe[0].to(device)
        return self._dev[device][:T]


class _Dropout(torch.autograd.Function):
    @staticmethod
    @_exact
    def forward(ctx, x, p):
        y, mask = _drop(x.contiguous(), p, True, _seed(True, p))
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    @_exact
    def backward(ctx, dy):
        return _drop_bwd(dy.contiguous(), ctx.saved_tensors[0], ctx.p), None


class ConformerEncoder(nn.Module):
    def __init__(self, d_model, N, heads, dropout):
        super().__init__()
        self.N, self.heads = N, heads
        self.pe = RelativePositionalEncoder(d_model, dropout=dropout)
        self.layers = nn.Sequential(*[ConformerEncoderLayer(d_model, heads, dropout) for _ in range(N)])

    def forward(self, src, keep):
        T = src.shape[1]
        if T > 1024:
            raise RuntimeError("asr_amd ConformerEncoder: more than 1024 subsampled frames (softmax kernel limit)")
        pe = self.pe.table(T, src.device)
        x = src
        if self.training and self.pe.p > 0.0:
            x, pe = _Dropout.apply(x, self.pe.p), _Dropout.apply(pe.contiguous(), self.pe.p)
        for layer in self.layers:
            x = layer(x, pe, keep)
        return x


class CNN_embedding(nn.Module):
    def __init__(self, idim, channels, odim):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, channels, 3, 2), nn.ReLU(), nn.Conv2d(channels, channels, 3, 2), nn.ReLU())
        self.out = nn.Sequential(nn.Linear(channels * (((idim - 1) // 2 - 1) // 2), odim, bias=False))

    def forward(self, x, x_mask):
        c = self.conv
        y = _CnnEmbeddingFunction.apply(x, c[0].weight, c[0].bias, c[2].weight, c[2].bias, self.out[0].weight)
        if x_mask is None:
            return y, None
        return y, x_mask[:, :, :-2:2][:, :, :-2:2]


class TransformerEncoder(nn.Module):
    """Transformer_encoder.py:441-451."""

    def __init__(self, src_vocab, d_model, N_e, heads, dropout):
        super().__init__()
        self.cnn_encoder = CNN_embedding(src_vocab, d_model, d_model)
        self.encoder = ConformerEncoder(d_model, N_e, heads, dropout)

    def forward(self, src, src_pos):
        if not src.is_cuda:
            raise RuntimeError("asr_amd.TransformerEncoder: input must be on the GPU; the HIP path has no CPU fallback")
        src_mask = (src_pos != 0).unsqueeze(-2)
        x, mask = self.cnn_encoder(src, src_mask)
        keep = mask[:, 0, :].to(device=src.device, dtype=torch.uint8).contiguous()
        return self.encoder(x, keep)
