"""Thin torch-tensor wrappers over the C ABI (pointers + sizes only cross it).

PyTorch is plumbing here: it owns device memory and the stream.  Every
function enqueues HIP kernels of libasr_amd.so on the current stream and
returns without synchronising.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("asr_amd: tensors must live on the GPU (no CPU path exists)")
    if t.dtype != dtype:
        raise RuntimeError(f"asr_amd: expected {dtype}, got {t.dtype}")
    return t


def _span(t, off, rows, inner, dd, s0, s1, nbatch, batch, batch_d, batch1, listed):
    """Bytes readable from element `off` of tensor t if all the view's corners stay inside t, else 0."""
    if not t.is_contiguous():
        return 0
    if listed:                      # rows named by a device list: any row of the tensor, single-level stride
        return (t.numel() - off) * 4 if dd == 0 and nbatch == 1 else 0
    r = rows - 1
    row_c = (0, r * s0) if dd == 0 else (0, min(r, dd - 1) * s0, (r // dd) * s1, (r // dd) * s1 + (r % dd) * s0,
                                         (r // dd) * s1 + (min(r, dd - 1)) * s0)
    n = nbatch - 1
    bat_c = (0,) if nbatch == 1 else ((0, n * batch) if batch_d == 0 else
                                      (0, min(n, batch_d - 1) * batch, (n // batch_d) * batch1,
                                       (n // batch_d) * batch1 + min(n, batch_d - 1) * batch))
    lo = min(row_c) + min(bat_c)
    hi = max(row_c) + max(bat_c) + inner
    return (t.numel() - off) * 4 if lo >= 0 and off + hi <= t.numel() else 0


def sgemm(A, B, Cm, M, N, K, *, a_mode=0, b_mode=0, a_stride=None, b_stride=None, c_stride=None,
          a_two=None, b_two=None, c_two=None, bias1=None, bias2=None, alpha=1.0, accumulate=False,
          act=0, splitk=1, nbatch=1, batch_a=0, batch_b=0, batch_c=0, a_off=0, b_off=0, c_off=0,
          m_idx=None, k_idx=None, bias_two=None, keep=None, keep_scale=1.0, batch_two=None, pad_ok=0,
          a_packed=False, b_packed=False, b_pk=None):
    """C[m,n] (+)= alpha * sum_k A(m,k) B(n,k) (+ biases); see include/asr_amd.h.

    *_stride: single row stride (elements); *_two = (d, s0, s1): two-level row
    offset (i // d) * s1 + (i % d) * s0.  *_off: element offsets into the tensors.
    m_idx / k_idx: int32 device row lists (M resp. K then count list entries).
    bias_two = (d, s1_bias1, s1_bias2): two-level bias index.  keep / keep_scale: uint8 mask laid
    out like C, applied to the stored value (fused dropout backward).
    b_pk = (packed tensor, row stride): a pre-split copy of the WHOLE B operand (pack_hilo; single-level rows, same
    b_mode), used instead of B when the library has a kernel for it at this shape (packed_ok, buffer loads possible),
    silently ignored otherwise -- the result is bit-identical either way.
    """
    lib = _lib.load()
    d = GemmDesc()
    es = 4
    d.A = A.data_ptr() + a_off * es
    d.B = B.data_ptr() + b_off * es
    d.C = Cm.data_ptr() + c_off * es
    d.M, d.N, d.K = M, N, K
    d.a_mode, d.b_mode = a_mode, b_mode
    for name, two, stride in (("a", a_two, a_stride), ("b", b_two, b_stride), ("c", c_two, c_stride)):
        if two is not None:
            dd, s0, s1 = two
        else:
            dd, s0, s1 = 0, stride, 0
        setattr(d, name + "_d", dd)
        setattr(d, name + "_s0", s0)
        setattr(d, name + "_s1", s1)
    d.bias1 = 0 if bias1 is None else bias1.data_ptr()
    d.bias2 = 0 if bias2 is None else bias2.data_ptr()
    d.alpha = alpha
    d.accumulate = int(accumulate)
    d.act = act
    d.splitk = splitk
    d.nbatch = nbatch
    d.batch_a, d.batch_b, d.batch_c = batch_a, batch_b, batch_c
    d.bias_d, d.bias1_s1, d.bias2_s1 = (0, 0, 0) if bias_two is None else bias_two
    d.c_keep = 0 if keep is None else keep.data_ptr()
    d.c_keep_scale = keep_scale
    # batch_two = (batch_d, a1, b1, c1): two-level batch index, see include/asr_amd.h
    d.batch_d, d.batch_a1, d.batch_b1, d.batch_c1 = (0, 0, 0, 0) if batch_two is None else batch_two
    d.pad_ok = pad_ok
    d.a_packed, d.b_packed = int(a_packed), int(b_packed)
    # readable span of A and B from their pointers, when every element the views can name lies inside the tensor
    # (a two-level view may step into ANOTHER allocation, e.g. the reverse direction's weights: then 0 = unknown)
    bd, ba1, bb1 = (d.batch_d, d.batch_a1, d.batch_b1) if batch_two is not None else (0, 0, 0)
    d.a_span = _span(A, a_off, M if a_mode == 0 else K, K if a_mode == 0 else M, d.a_d, d.a_s0, d.a_s1,
                     nbatch, batch_a, bd, ba1, m_idx is not None or (k_idx is not None and a_mode == 1))
    d.b_span = _span(B, b_off, N if b_mode == 0 else K, K if b_mode == 0 else N, d.b_d, d.b_s0, d.b_s1,
                     nbatch, batch_b, bd, bb1, k_idx is not None and b_mode == 1)
    if (b_pk is not None and (not a_packed or b_mode == 0) and a_mode == 0 and k_idx is None and nbatch == 1
            and splitk == 1 and d.a_span > 0 and d.a_span < 0x7ffffff0 and packed_ok(M, N)):
        Bp, bps = b_pk
        span = _span(Bp, 0, N if b_mode == 0 else K, K if b_mode == 0 else N, 0, bps, 0, 1, 0, 0, 0, False)
        if span > 0 and span < 0x7ffffff0 and (b_mode == 0 or K * bps * 4 < 0x7ffffff0):
            d.B, d.b_d, d.b_s0, d.b_s1, d.b_span, d.b_packed = Bp.data_ptr(), 0, bps, 0, span, 1
    if a_packed and not d.b_packed:
        raise RuntimeError("asr_amd.ops.sgemm: a pre-split A operand needs a pre-split B (b_pk) at a shape packed_ok accepts")
    for name, idx, n in (("m_idx", m_idx, M), ("k_idx", k_idx, K)):
        if idx is None:
            setattr(d, name, 0)
            continue
        if not idx.is_cuda or idx.dtype != torch.int32 or not idx.is_contiguous() or idx.numel() < n:
            raise RuntimeError(f"asr_amd.sgemm: {name} must be a contiguous int32 device tensor of >= {n} entries")
        setattr(d, name, idx.data_ptr())
    check(lib.asr_sgemm(C.byref(d), _stream()))


GEMM_MODES = {"f32": 0, "bf16x3": 1, "bf16": 2}


def packed_ok(M, N):
    """True when asr_sgemm has a pre-split-operand kernel for an (M x N) product with a k-contiguous A: the bf16x3
    mode, 128 x 128 tiles (asr_sgemm picks them from 320 tiles on) and no tile override in the environment."""
    import os
    return (_lib.load().asr_gemm_precision(-1) == 1 and M >= 128 and N >= 128
            and ((M + 127) // 128) * ((N + 127) // 128) >= 320 and not os.environ.get("ASR_GEMM_BF_TILE")
            and os.environ.get("ASR_GEMM_BUF", "1") != "0" and os.environ.get("ASR_GEMM_PACKED", "1") != "0")


def packed_dw_ok(M, N, K, splitk):
    """True when asr_sgemm has a kernel for a weight-gradient product (a_mode 1, b_mode 1) whose B operand is
    pre-split: bf16x3, 128 x 128 tiles (asr_sgemm picks them from 320 tiles x splits on, from 256 when K is split)."""
    import os
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * max(1, min(splitk, (K + 31) // 32))
    return (_lib.load().asr_gemm_precision(-1) == 1 and M >= 128 and N >= 128 and K >= 32
            and (tiles >= 320 or (splitk > 1 and tiles >= 256)) and not os.environ.get("ASR_GEMM_BF_TILE")
            and os.environ.get("ASR_GEMM_BUF", "1") != "0" and os.environ.get("ASR_GEMM_PACKED", "1") != "0")


def dropout_fwd_packed(x, p, seed):
    """Inverted dropout of x with the output as {hi:lo} words (in a float32 tensor of x's shape) and the uint8 keep mask:
    asr_dropout_fwd's mask and values for the same p and seed, pre-split for asr_sgemm (a_packed / b_packed)."""
    x = _req(x).contiguous()
    y = torch.empty_like(x)
    mask = torch.empty(x.numel(), device=x.device, dtype=torch.uint8)
    check(_lib.load().asr_dropout_fwd_packed(_p(x), _p(y), _p(mask), x.numel(), C.c_float(p), C.c_uint64(seed), _stream()))
    return y, mask


def pack_hilo(src0, src1=None):
    """{bf16 hi : bf16 lo} words (stored in a float32 tensor) of src0, followed by src1's when given: the
    pre-split operand form of asr_sgemm (b_packed)."""
    src0 = _req(src0).contiguous()
    n = src0.numel()
    if src1 is not None:
        src1 = _req(src1).contiguous()
        assert src1.numel() == n
    out = torch.empty((2 if src1 is not None else 1) * n, device=src0.device, dtype=torch.float32)
    check(_lib.load().asr_pack_hilo(_p(src0), _p(src1), n, _p(out), _stream()))
    return out


class gemm_precision:
    """with gemm_precision("f32" | "bf16x3" | "bf16"): ... -- arithmetic of the asr_sgemm products enqueued inside
    the block (include/asr_amd.h, asr_gemm_precision); the previous mode comes back on exit.  The mode is read on
    the host when a product is enqueued, so the block may end before the products have run."""

    def __init__(self, mode):
        self.mode = GEMM_MODES[mode] if isinstance(mode, str) else int(mode)

    def __enter__(self):
        self.old = _lib.load().asr_gemm_precision(self.mode)
        return self

    def __exit__(self, *exc):
        _lib.load().asr_gemm_precision(self.old)
        return False


def elem_delta(a, b):
    """Element distance from tensor a to tensor b (both fp32) if a two-level GEMM view can
    address them as one operand (16-byte aligned distance), else None."""
    diff = b.data_ptr() - a.data_ptr()
    return diff // 4 if diff % 16 == 0 else None


def pick_splitk(M, N, K, nbatch=1):
    """Split K over workgroups (weight gradients: small M x N, K = all frames of the batch).  With 8 or more 128x128
    tiles: as many splits as keep the launch within ONE resident round of 512 workgroups (2 per CU) -- a second,
    partly filled round costs more than the longer K loops, and every split adds an M x N pass of float atomics
    (tools/gemm_splitk.py).  Fewer tiles: 64x64 tiles, about three workgroups per CU."""
    if M >= 128 and N >= 128:
        c128 = ((M + 127) // 128) * ((N + 127) // 128) * nbatch
        if c128 >= 8:
            return int(max(1, min(512 // c128, K // 512, 64)))
    if M < 128 or N < 128:
        tiles = ((M + 63) // 64) * ((N + 63) // 64) * nbatch
    else:
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * nbatch
    if tiles >= 256 or K < 512:
        return 1
    want = (768 + tiles - 1) // tiles
    return int(max(1, min(want, K // 512, 64)))


def linear_nt(x2d, W, out2d, bias1=None, bias2=None, act=0, accumulate=False):
    """out = x2d @ W^T (+bias): x2d (M,K) and W (N,K) row-major contiguous."""
    M, K = x2d.shape
    N = W.shape[0]
    sgemm(x2d, W, out2d, M, N, K, a_stride=x2d.stride(0), b_stride=W.stride(0), c_stride=out2d.stride(0),
          bias1=bias1, bias2=bias2, act=act, accumulate=accumulate)


def matmul_nn(x2d, W, out2d, accumulate=False):
    """out = x2d @ W: x2d (M,K), W (K,N) row-major."""
    M, K = x2d.shape
    N = W.shape[1]
    sgemm(x2d, W, out2d, M, N, K, a_stride=x2d.stride(0), b_mode=1, b_stride=W.stride(0),
          c_stride=out2d.stride(0), accumulate=accumulate)


def matmul_tn(a2d, b2d, out2d, zeroed=False):
    """out (M,N) = a2d^T @ b2d with a2d (K,M), b2d (K,N) row-major (weight gradients).

    K is split over workgroups when M x N alone cannot fill the chip; out is
    zeroed first because the split-K partials are summed with float atomics.
    """
    K, M = a2d.shape
    N = b2d.shape[1]
    sk = pick_splitk(M, N, K)
    if sk > 1 and not zeroed:          # zeroed: the caller cleared out2d already (one fill for several gradients)
        out2d.zero_()
    sgemm(a2d, b2d, out2d, M, N, K, a_mode=1, a_stride=a2d.stride(0), b_mode=1, b_stride=b2d.stride(0),
          c_stride=out2d.stride(0), splitk=sk)


def colsum(a2d, out, accumulate=False):
    lib = _lib.load()
    M, N = a2d.shape
    nws = int(lib.asr_colsum_ws_floats(M, N))
    part = torch.empty(nws, device=a2d.device, dtype=torch.float32)
    check(lib.asr_colsum(_p(a2d), M, N, a2d.stride(0), 0, 0, _p(out), int(accumulate), _p(part), _stream()))


def colsum_multi(pairs):
    """pairs: [(a2d (M, n_i) contiguous, out (n_i,))...], all of the same height M, at most 8: two launches."""
    lib = _lib.load()
    l = _lib.ColsumList()
    l.count = len(pairs)
    M = pairs[0][0].shape[0]
    for i, (a, out) in enumerate(pairs):
        if a.shape[0] != M or not a.is_contiguous() or out.numel() != a.shape[1]:
            raise RuntimeError("asr_amd.colsum_multi: matrices must be contiguous, of equal height, out of width n")
        l.src[i], l.out[i], l.n[i] = a.data_ptr(), out.data_ptr(), a.shape[1]
    part = torch.empty(int(lib.asr_colsum_multi_ws_floats(C.byref(l), M)), device=pairs[0][0].device, dtype=torch.float32)
    check(lib.asr_colsum_multi(C.byref(l), M, _p(part), _stream()))


def transpose(x2d):
    """Contiguous (R,C) -> new contiguous (C,R); leading dims are batched."""
    lib = _lib.load()
    *lead, R, Cc = x2d.shape
    nb = 1
    for v in lead:
        nb *= v
    out = torch.empty(*lead, Cc, R, device=x2d.device, dtype=torch.float32)
    check(lib.asr_transpose(_p(x2d), _p(out), R, Cc, nb, _stream()))
    return out


def scatter_rows(src2d, idx, table):
    lib = _lib.load()
    n, w = src2d.shape
    check(lib.asr_scatter_rows(_p(src2d), _p(idx), n, w, src2d.stride(0), _p(table), table.shape[0], _stream()))


def mul_dtanh(dU, U):
    lib = _lib.load()
    out = torch.empty_like(U)
    check(lib.asr_mul_dtanh(_p(dU), _p(U), _p(out), U.numel(), _stream()))
    return out


def scale_(x, s):
    check(_lib.load().asr_scale(_p(x), x.numel(), float(s), _stream()))
    return x
