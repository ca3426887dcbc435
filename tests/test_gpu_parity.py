"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden
vectors and against the CPU oracle on seeded inputs.

Tolerances are the north star's fp32 bar: loss rel-err <= 1e-4, gradients
<= 1e-3 (scale-relative max error, tests/_util.rel_err); youtput <= 1e-4.
Greedy decoding must be token-identical.
"""
import copy

import numpy as np
import pytest
import torch

from tests._util import (CASES, LM_CASES, LM_RUNS, V2_CASES, V2_RUNS, load_case, load_lm, load_v2, params_of, rel_err,
                         greedy_raw_from_table)

pytestmark = pytest.mark.gpu

LOSS_TOL, OUT_TOL, GRAD_TOL = 1e-4, 1e-4, 1e-3


def _lib_prec():
    from asr_amd import _lib
    return int(_lib.load().asr_gemm_precision(-1))


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU: refusing to pass on a fallback"
    return torch.device("cuda:0")


def _build(hp, state):
    from asr_amd.Models.AttModel import AttModel
    m = AttModel(hp)
    m.load_state_dict(state)
    return m.to(_dev()).train()


# ------------------------------------------------------------------ sgemm --
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 7, 3), (64, 64, 16), (130, 70, 33), (257, 129, 100),
                                   (300, 260, 64), (1000, 384, 40), (2304, 1408, 50), (2304, 1408, 64),
                                   (11968, 256, 256)])
@pytest.mark.parametrize("a_mode,b_mode", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_sgemm_modes(M, N, K, a_mode, b_mode, gemm_mode):
    from asr_amd import ops
    tol, _ = gemm_mode
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + a_mode * 2 + b_mode)
    A = torch.randn((M, K) if a_mode == 0 else (K, M), generator=g)
    Bm = torch.randn((N, K) if b_mode == 0 else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    want = (A if a_mode == 0 else A.t()).double() @ (Bm if b_mode == 0 else Bm.t()).double().t() + bias.double()
    Ad, Bd, bd = A.to(dev), Bm.to(dev), bias.to(dev)
    out = torch.full((M, N), 7.0, device=dev)
    ops.sgemm(Ad, Bd, out, M, N, K, a_mode=a_mode, b_mode=b_mode, a_stride=A.shape[1], b_stride=Bm.shape[1],
              c_stride=N, bias1=bd)
    assert rel_err(out.cpu(), want) < tol
    # accumulate + split-K (atomic) variants
    out2 = torch.ones(M, N, device=dev)
    ops.sgemm(Ad, Bd, out2, M, N, K, a_mode=a_mode, b_mode=b_mode, a_stride=A.shape[1], b_stride=Bm.shape[1],
              c_stride=N, bias1=bd, accumulate=True)
    assert rel_err(out2.cpu(), want + 1.0) < tol
    out3 = torch.zeros(M, N, device=dev)
    ops.sgemm(Ad, Bd, out3, M, N, K, a_mode=a_mode, b_mode=b_mode, a_stride=A.shape[1], b_stride=Bm.shape[1],
              c_stride=N, bias1=bd, splitk=3)
    assert rel_err(out3.cpu(), want) < tol


def test_sgemm_two_level_and_batched():
    from asr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, T, D, N = 3, 11, 8, 20
    x = torch.randn(B, T, D, generator=g)
    W = torch.randn(N, D, generator=g)
    out = torch.empty(T, B, N, device=dev)
    ops.sgemm(x.to(dev), W.to(dev), out, T * B, N, D, a_two=(B, T * D, D), b_stride=D, c_stride=N)
    want = torch.einsum("btd,nd->tbn", x.double(), W.double())
    assert rel_err(out.cpu(), want) < 2e-6
    # batched TN with accumulate: dh[b] += alpha[:, b]^T dg[:, b]
    L, E = 6, 12
    al = torch.randn(L, B, T, generator=g)
    dg = torch.randn(L, B, E, generator=g)
    dh0 = torch.randn(B, T, E, generator=g)
    dh = dh0.clone().to(dev)
    ops.sgemm(al.to(dev), dg.to(dev), dh, T, E, L, a_mode=1, a_stride=B * T, b_mode=1, b_stride=B * E,
              c_stride=E, accumulate=True, nbatch=B, batch_a=T, batch_b=E, batch_c=T * E)
    want = dh0.double() + torch.einsum("lbt,lbe->bte", al.double(), dg.double())
    assert rel_err(dh.cpu(), want) < 2e-6


@pytest.mark.parametrize("R,N,K,keep", [(9, 5, 3, 0.6), (300, 130, 40, 0.8), (2000, 256, 64, 0.75),
                                        (17000, 1280, 240, 0.8), (40, 12, 8, 1.0)])
def test_sgemm_row_lists(R, N, K, keep, gemm_mode):
    """m_idx / k_idx: the products over a list of rows equal the dense products over those rows,
    unlisted rows of C stay untouched (a ragged batch skipping its padded frames)."""
    from asr_amd import ops
    tol, _ = gemm_mode
    dev = _dev()
    g = torch.Generator().manual_seed(R + N + K)
    rows = torch.nonzero(torch.rand(R, generator=g) < keep).flatten().to(torch.int32)
    if keep == 1.0:
        rows = torch.arange(R, dtype=torch.int32)
    rows = rows[torch.randperm(rows.numel(), generator=g)] if R < 100 else rows     # any order is allowed
    n = rows.numel()
    X = torch.randn(R, K, generator=g)
    W = torch.randn(N, K, generator=g)
    idx = rows.to(dev)
    Xd, Wd = X.to(dev), W.to(dev)
    for b_mode, Wop in ((0, Wd), (1, Wd.t().contiguous())):
        out = torch.full((R, N), 7.0, device=dev)
        ops.sgemm(Xd, Wop, out, n, N, K, b_mode=b_mode, a_stride=K, b_stride=Wop.shape[1], c_stride=N, m_idx=idx)
        want = torch.full((R, N), 7.0, dtype=torch.float64)
        want[rows.long()] = X.double()[rows.long()] @ W.double().t()
        assert rel_err(out.cpu(), want) < tol
    # weight-gradient form: dW (N x K) = sum over listed rows of dY[r]^T X[r]
    dY = torch.randn(R, N, generator=g)
    dYd = dY.to(dev)
    want = dY.double()[rows.long()].t() @ X.double()[rows.long()]
    for sk in (1, 3):
        gW = torch.zeros(N, K, device=dev)
        ops.sgemm(dYd, Xd, gW, N, K, n, a_mode=1, a_stride=N, b_mode=1, b_stride=K, c_stride=K, splitk=sk, k_idx=idx)
        assert rel_err(gW.cpu(), want) < 5 * tol  # fp32 sums over up to 13 600 rows
    with pytest.raises(RuntimeError):
        ops.sgemm(dYd, Xd, gW, N, K, n, a_mode=1, a_stride=N, b_mode=1, b_stride=K, c_stride=K, m_idx=idx)
    with pytest.raises(RuntimeError):
        ops.sgemm(Xd, Wd, out, n, N, K, a_stride=K, b_stride=K, c_stride=N, k_idx=idx)


def test_dropout_packed_output_and_presplit_weight_gradient():
    """asr_dropout_fwd_packed writes asr_dropout_fwd's mask and values as {hi:lo} words (bit for bit what asr_pack_hilo
    makes of the fp32 output), and the weight-gradient product dW = dG^T X reads X in that form (b_packed) -- with the
    k list of a ragged batch (pointer loaders) and without (buffer loads) -- giving the fp32-operand product's result
    up to the order of the split-K atomics; as the A operand of the input projection it is bit-identical."""
    from asr_amd import _lib, ops
    import ctypes as C
    lib = _lib.load()
    if lib.asr_gemm_precision(-1) != 1:
        pytest.skip("the pre-split kernels belong to the bf16x3 mode (ASR_GEMM_PRECISION selects another one here)")
    dev = _dev()
    T, B, D, H8 = 400, 32, 640, 2560
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, D, generator=g).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    y = torch.empty_like(x)
    m = torch.empty(x.numel(), device=dev, dtype=torch.uint8)
    assert lib.asr_dropout_fwd(x.data_ptr(), y.data_ptr(), m.data_ptr(), x.numel(), C.c_float(0.2), C.c_uint64(99), st) == 0
    yp, mp = ops.dropout_fwd_packed(x, 0.2, 99)
    assert torch.equal(m, mp)
    assert torch.equal(yp.view(torch.int32), ops.pack_hilo(y).view_as(y).view(torch.int32))
    dG = torch.randn(T * B, H8, generator=g).to(dev)
    rows = torch.arange(0, T * B, dtype=torch.int32)[torch.rand(T * B, generator=g) < 0.8].to(dev)
    X2, Xp2 = y.view(T * B, D), yp.view(T * B, D)
    for kl, K in ((rows, rows.numel()), (None, T * B)):
        sk = ops.pick_splitk(H8, D, K)
        assert sk > 1 and ops.packed_dw_ok(H8, D, K, sk)
        outs = []
        for Bm, pk in ((X2, False), (Xp2, True)):
            out = torch.zeros(H8, D, device=dev)
            ops.sgemm(dG, Bm, out, H8, D, K, a_mode=1, a_stride=H8, b_mode=1, b_stride=D, c_stride=D, splitk=sk,
                      k_idx=kl, b_packed=pk)
            outs.append(out)
        scale = float(outs[0].abs().max())
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * scale
        sel = rows.long() if kl is not None else slice(None)
        want = dG.double()[sel].t() @ X2.double()[sel]
        assert rel_err(outs[1].cpu(), want.cpu()) < 2e-5
    # as the A operand of the projection (with pre-split weights): bit for bit the fp32-operand product
    W = torch.randn(H8, D, generator=g).to(dev)
    Wp = ops.pack_hilo(W).view_as(W)
    C1, C2 = torch.empty(T * B, H8, device=dev), torch.empty(T * B, H8, device=dev)
    ops.sgemm(X2, W, C1, rows.numel(), H8, D, a_stride=D, b_stride=D, c_stride=H8, m_idx=rows)
    ops.sgemm(Xp2, W, C2, rows.numel(), H8, D, a_stride=D, b_stride=D, c_stride=H8, m_idx=rows, b_pk=(Wp, D), a_packed=True)
    assert torch.equal(C1[rows.long()], C2[rows.long()])
    with pytest.raises(RuntimeError):          # a pre-split A alone has no kernel
        ops.sgemm(Xp2, W, C2, T * B, H8, D, a_stride=D, b_stride=D, c_stride=H8, a_packed=True)


def test_encoder_with_presplit_layer_inputs_matches_fp32_inputs():
    """Encoder step at the config-2 widths with the inter-layer dropout written pre-split (default where asr_sgemm has
    the kernels) against the same step with fp32 layer inputs: same dropout masks (same seed), bit-identical forward
    (the projections split the same two bf16 either way), gradients equal up to the split-K atomics' order."""
    from asr_amd.Models import encoder as E
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=3, encoder_dropout=0.2)
    torch.manual_seed(12)
    enc = E.Encoder(hp).to(dev).train()
    B, T = 32, 420
    lengths = torch.linspace(T, T * 3 // 5, B).long()
    x = torch.randn(B, T, 240)
    for b in range(B):
        x[b, lengths[b]:] = 0
    w = torch.randn(B, T, 640, device=dev)
    outs = []
    old = E.PACK_INPUT
    try:
        for flag in (False, True):
            E.PACK_INPUT = flag
            for q in enc.parameters():
                q.grad = None
            torch.manual_seed(77)
            h = enc(x.to(dev), lengths)
            (h * w).sum().backward()
            torch.cuda.synchronize()
            outs.append((h.detach().clone(), [q.grad.clone() for q in enc.parameters()]))
    finally:
        E.PACK_INPUT = old
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b_ in zip(outs[0][1], outs[1][1]):
        assert float((a - b_).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12
    assert not E.persistent_error_flags()


@pytest.mark.parametrize("M,N,K,b_mode", [(12800, 2560, 640, 0), (12800, 640, 2560, 1), (56000, 128, 96, 0)])
def test_sgemm_presplit_weights_bit_identical(M, N, K, b_mode):
    """asr_sgemm with the B operand handed over pre-split (asr_pack_hilo: one {bf16 hi : bf16 lo} word per element,
    the split done once instead of by every row tile) against the same product splitting fp32 operands on the fly:
    the arithmetic is the same, so the results are bit for bit equal -- with a row list and a bias as the encoder's
    projection uses them, and with both operands pre-split.  In exact-fp32 mode the packed copy is ignored."""
    from asr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) if b_mode == 0 else torch.randn(K, N, generator=g)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    rows = torch.arange(0, M, 1, dtype=torch.int32)[torch.rand(M, generator=g) < 0.8].to(dev)
    Wp = ops.pack_hilo(W).view_as(W)
    w_words = Wp.view(torch.int32)
    w_bits = W.view(torch.int32)
    assert bool(((w_words >> 16) - (w_bits >> 16)).abs().max() <= 1)          # the hi half is bf16(x): top 16 bits, rounded
    bs = K if b_mode == 0 else N
    if _lib_prec() != 1:
        pytest.skip("the pre-split kernels belong to the bf16x3 mode (ASR_GEMM_PRECISION selects another one here)")
    assert ops.packed_ok(rows.numel(), N)
    outs = []
    for pk in (None, (Wp, bs)):
        C_ = torch.full((M, N), 7.0, device=dev)
        ops.sgemm(A, W, C_, rows.numel(), N, K, a_stride=K, b_mode=b_mode, b_stride=bs, c_stride=N, bias1=bias,
                  m_idx=rows, b_pk=pk)
        outs.append(C_)
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0][rows.long()] - 7.0).abs().max()) > 0 and bool((outs[0] == 7.0).any())     # unlisted rows untouched
    want = A.double()[rows.long()] @ (W.double().t() if b_mode == 0 else W.double()) + bias.double()
    assert rel_err(outs[1][rows.long()].cpu(), want.cpu()) < 2e-5
    if b_mode == 0:
        Ap = ops.pack_hilo(A).view_as(A)
        C2 = torch.empty(M, N, device=dev)
        C3 = torch.empty(M, N, device=dev)
        ops.sgemm(Ap, Wp, C2, M, N, K, a_stride=K, b_stride=K, c_stride=N, a_packed=True, b_packed=True)
        ops.sgemm(A, W, C3, M, N, K, a_stride=K, b_stride=K, c_stride=N)
        assert torch.equal(C2, C3)
    with ops.gemm_precision("f32"):          # no pre-split kernel in this mode: the fp32 operand is used
        C4 = torch.empty(M, N, device=dev)
        ops.sgemm(A, W, C4, M, N, K, a_stride=K, b_mode=b_mode, b_stride=bs, c_stride=N, b_pk=(Wp, bs))
        want = A.double() @ (W.double().t() if b_mode == 0 else W.double())
        assert rel_err(C4.cpu(), want.cpu()) < 5e-6          # fp32 accumulation over up to 2560 terms



def test_sgemm_two_tensor_operand_and_fused_keep(gemm_mode):
    """One launch over two separately allocated weight / bias tensors (two-level views), and the
    dropout-backward mask applied in the epilogue."""
    from asr_amd import ops
    tol, _ = gemm_mode
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    M, K, Nh = 300, 48, 64
    X = torch.randn(M, K, generator=g)
    Wf, Wr = torch.randn(Nh, K, generator=g), torch.randn(Nh, K, generator=g)
    bf1, br1 = torch.randn(Nh, generator=g), torch.randn(Nh, generator=g)
    bf2, br2 = torch.randn(Nh, generator=g), torch.randn(Nh, generator=g)
    junk = torch.empty(12345, device=dev)                       # keeps the two allocations apart
    Wfd, Wrd, b1f, b1r, b2f, b2r = (t.to(dev) for t in (Wf, Wr, bf1, br1, bf2, br2))
    dw, d1, d2 = ops.elem_delta(Wfd, Wrd), ops.elem_delta(b1f, b1r), ops.elem_delta(b2f, b2r)
    assert None not in (dw, d1, d2)
    out = torch.empty(M, 2 * Nh, device=dev)
    ops.sgemm(X.to(dev), Wfd, out, M, 2 * Nh, K, a_stride=K, b_two=(Nh, K, dw), c_stride=2 * Nh,
              bias1=b1f, bias2=b2f, bias_two=(Nh, d1, d2))
    want = torch.cat([X.double() @ Wf.double().t() + bf1.double() + bf2.double(),
                      X.double() @ Wr.double().t() + br1.double() + br2.double()], 1)
    assert rel_err(out.cpu(), want) < tol
    # dX form: reduction over the rows of both weight tensors, mask in the epilogue
    dG = torch.randn(M, 2 * Nh, generator=g)
    keep = (torch.rand(M, K, generator=g) < 0.8).to(torch.uint8)
    dX = torch.empty(M, K, device=dev)
    ops.sgemm(dG.to(dev), Wfd, dX, M, K, 2 * Nh, a_stride=2 * Nh, b_mode=1, b_two=(Nh, K, dw), c_stride=K,
              keep=keep.to(dev), keep_scale=1.25)
    want = (dG.double() @ torch.cat([Wf, Wr], 0).double()) * keep.double() * 1.25
    assert rel_err(dX.cpu(), want) < tol
    assert torch.equal(dX.cpu() == 0, keep == 0) or float((dX.cpu() == 0).sum()) >= float((keep == 0).sum())
    with pytest.raises(RuntimeError):
        ops.sgemm(dG.to(dev), Wfd, dX, M, K, 2 * Nh, a_stride=2 * Nh, b_mode=1, b_two=(Nh, K, dw), c_stride=K,
                  keep=keep.to(dev), splitk=2)
    del junk


def test_helpers():
    from asr_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    a = torch.randn(1300, 70, generator=g)
    out = torch.empty(70, device=dev)
    ops.colsum(a.to(dev), out)
    assert rel_err(out.cpu(), a.double().sum(0)) < 1e-6
    x = torch.randn(2, 37, 50, generator=g)
    assert torch.equal(ops.transpose(x.to(dev)).cpu(), x.transpose(1, 2).contiguous())
    src = torch.randn(40, 24, generator=g)
    idx = torch.randint(0, 9, (40,), generator=g)
    table = torch.full((9, 24), 3.0, device=dev)
    ops.scatter_rows(src.to(dev), idx.to(dev), table)
    want = torch.zeros(9, 24, dtype=torch.float64).index_add_(0, idx, src.double())
    assert rel_err(table.cpu(), want) < 1e-6


# ------------------------------------------------------------------- loss --
@pytest.mark.parametrize("B,L,V,T_norm,B_norm", [(3, 5, 11, True, False), (4, 7, 1000, True, True),
                                                 (2, 9, 7000, False, False), (3, 4, 37, True, False),
                                                 (2, 3, 5003, True, False)])
def test_loss_kernel_vs_oracle(B, L, V, T_norm, B_norm):
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from oracle import ref_model as R
    dev = _dev()
    g = torch.Generator().manual_seed(V + L)
    z = (torch.randn(B, L, V, generator=g) * 3).requires_grad_(True)
    tgt = torch.randint(0, V, (B, L), generator=g)
    tl = torch.randint(1, L + 1, (B,), generator=g)
    want = R.label_smoothing_loss(z.double(), tgt, tl, T_norm, B_norm)
    gw, = torch.autograd.grad(want, z)
    zd = z.detach().to(dev).requires_grad_(True)
    got = label_smoothing_loss(zd, tgt.to(dev), tl.to(dev), T_norm, B_norm)
    got.backward()
    assert abs(float(got) - float(want)) / abs(float(want)) < 1e-5
    assert rel_err(zd.grad.cpu(), gw) < 1e-5


# ------------------------------------------------------ encoder / decoder --
@pytest.mark.parametrize("B,T,D,H,N,lens", [(3, 17, 12, 8, 2, [17, 9, 4]), (5, 33, 6, 12, 1, [20, 33, 1, 33, 7]),
                                            (18, 21, 10, 32, 3, None), (2, 40, 240, 320, 2, [40, 31]),
                                            (35, 9, 5, 8, 1, None), (2, 1, 4, 8, 2, [1, 1]),
                                            (33, 12, 6, 64, 2, None)])
def test_blstm_vs_oracle(B, T, D, H, N, lens, blstm_mode):
    from asr_amd.Models.encoder import Encoder
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=D, frame_stacking=1, num_hidden_nodes=H, num_encoder_layer=N, encoder_dropout=0.0)
    torch.manual_seed(B * T + H)
    ref = R.Encoder(hp).double()
    lengths = torch.tensor(lens) if lens is not None else torch.randint(1, T + 1, (B,))
    if lens is None:
        lengths[0] = T
    x = torch.randn(B, T, D)
    for b in range(B):
        x[b, lengths[b]:] = 0
    xr = x.double().requires_grad_(True)
    hr = ref(xr, lengths)
    w = torch.randn(B, T, 2 * H).double()
    (hr * w).sum().backward()
    enc = Encoder(hp)
    enc.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    enc = enc.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    hd = enc(xd, lengths)
    (hd * w.float().to(dev)).sum().backward()
    assert rel_err(hd.detach().cpu(), hr.detach()) < 2e-5
    for b in range(B):
        assert float(hd[b, lengths[b]:].abs().max() if lengths[b] < T else 0.0) == 0.0
    assert rel_err(xd.grad.cpu(), xr.grad) < GRAD_TOL
    for (k, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < GRAD_TOL, k
    from asr_amd.Models.encoder import persistent_error_flags
    assert not persistent_error_flags()


@pytest.mark.parametrize("B,T,H", [(32, 60, 320), (20, 33, 64), (32, 17, 512)])
def test_blstm_xcd_grouped_grid_is_bit_identical(B, T, H):
    """asr_blstm_xcd_groups: one exchange group per XCD label (hand-off through that XCD's L2 with plain stores once the
    group has verified its placement) against the three-dimensional grid with write-through stores, through the C ABI:
    same arithmetic in the same order, so G / Cs / Hout of the forward and dG of the backward must be bit-identical."""
    from asr_amd import _lib
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + T + H)
    W = (torch.randn(2, 4 * H, H, generator=g) * 0.08).to(dev)
    WT = W.transpose(1, 2).contiguous()
    G0 = torch.randn(T, B, 2, 4 * H, generator=g).to(dev)
    dH = torch.randn(T, B, 2 * H, generator=g).to(dev)
    lengths = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32)
    lengths[0] = T
    lengths = lengths.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    old = lib.asr_blstm_xcd_groups(-1)
    try:
        for mode in (0, 1):
            lib.asr_blstm_xcd_groups(mode)
            ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev)
            Gw, Cs, Hout = G0.clone(), torch.zeros(T, B, 2, H, device=dev), torch.zeros(T, B, 2 * H, device=dev)
            dc = torch.zeros(2 * B * H, device=dev)
            check = _lib.check
            check(lib.asr_blstm_fwd(W.data_ptr(), Gw.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H,
                                    lengths.data_ptr(), T, B, H, ws.data_ptr(), st))
            fwd = (Gw.clone(), Cs.clone(), Hout.clone())
            check(lib.asr_blstm_bwd(W.data_ptr(), WT.data_ptr(), Gw.data_ptr(), Cs.data_ptr(), dH.data_ptr(), B * 2 * H,
                                    2 * H, lengths.data_ptr(), dc.data_ptr(), T, B, H, ws.data_ptr(), st))
            torch.cuda.synchronize()
            assert int(ws.view(torch.int32)[1024]) == 0          # no in-launch wait timed out
            outs.append(fwd + (Gw.clone(),))
    finally:
        lib.asr_blstm_xcd_groups(old)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[1][3]).all() and float(outs[1][2].abs().max()) > 0


def test_persistent_and_stepwise_recurrence_agree(blstm_mode):
    """The one-launch-per-layer (in-launch hand-off) kernels against the
    one-launch-per-time-step kernels at the production width, ragged batch of 32;
    also checks that no in-launch wait ever timed out."""
    from asr_amd.Models import encoder as E
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=40, frame_stacking=1, num_hidden_nodes=320, num_encoder_layer=2, encoder_dropout=0.0)
    torch.manual_seed(77)
    enc = E.Encoder(hp).to(dev).train()
    B, T = 32, 70
    lengths = torch.linspace(T, 20, B).long()
    x = torch.randn(B, T, 40)
    for b in range(B):
        x[b, lengths[b]:] = 0
    w = torch.randn(B, T, 640, device=dev)
    outs = []
    for persistent in (True, False):
        E.PERSISTENT = persistent
        try:
            for p in enc.parameters():
                p.grad = None
            xd = x.to(dev).requires_grad_(True)
            h = enc(xd, lengths)
            (h * w).sum().backward()
            outs.append((h.detach().cpu(), xd.grad.cpu(), [p.grad.cpu() for p in enc.parameters()]))
        finally:
            E.PERSISTENT = True
    (h1, dx1, g1), (h2, dx2, g2) = outs
    # the persistent kernels carry a 1-bit epoch tag in the mantissa LSB of the exchanged
    # copies of h / the partial products: agreement is to a few ulp, not bitwise
    assert rel_err(h1, h2) < 5e-6
    assert rel_err(dx1, dx2) < 1e-5
    for a, b_ in zip(g1, g2):
        assert rel_err(a, b_) < 1e-5
    assert not E.persistent_error_flags()


@pytest.mark.parametrize("H,B,T,chunks,p", [(320, 32, 70, 4, 0.0), (320, 32, 131, 3, 0.3), (64, 7, 97, 5, 0.3),
                                            (10, 5, 66, 4, 0.3), (320, 20, 64, 64, 0.0)])
def test_chunk_pipelined_encoder_matches_whole_layer_launches(H, B, T, chunks, p):
    """The chunked recurrence launches (asr_blstm_fwd_range / asr_blstm_bwd_range) with the next
    layer's projection, the dropout blocks and dX accumulated per chunk on the second stream,
    against one launch per layer with whole-tensor GEMMs: same dropout masks (same seed), results
    equal up to the summation order of the split GEMMs.  H=10 takes the step-wise kernels."""
    from asr_amd.Models import encoder as E
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=24, frame_stacking=1, num_hidden_nodes=H, num_encoder_layer=3, encoder_dropout=p)
    torch.manual_seed(H + T)
    enc = E.Encoder(hp).to(dev).train()
    lengths = torch.linspace(T, max(1, T // 3), B).long()
    x = torch.randn(B, T, 24)
    for b in range(B):
        x[b, lengths[b]:] = 0
    w = torch.randn(B, T, 2 * H, device=dev)
    outs = []
    old = E.PIPE_CHUNKS
    for nch in (1, chunks):
        E.PIPE_CHUNKS = nch
        try:
            for q in enc.parameters():
                q.grad = None
            torch.manual_seed(4242)            # the dropout seed is drawn from torch's generator
            xd = x.to(dev).requires_grad_(True)
            h = enc(xd, lengths)
            (h * w).sum().backward()
            torch.cuda.synchronize()
            outs.append((h.detach().cpu(), xd.grad.cpu(), [q.grad.cpu() for q in enc.parameters()]))
        finally:
            E.PIPE_CHUNKS = old
    (h1, dx1, g1), (h2, dx2, g2) = outs
    assert rel_err(h1, h2) < 1e-5
    assert rel_err(dx1, dx2) < 2e-5
    for a, b_ in zip(g1, g2):
        assert rel_err(a, b_) < 2e-5
    for b in range(B):
        if lengths[b] < T:
            assert float(h2[b, lengths[b]:].abs().max()) == 0.0
    assert not E.persistent_error_flags()


@pytest.mark.parametrize("B,T,He,Hd,V,L,lens", [(3, 17, 8, 8, 11, 5, [17, 9, 4]), (4, 70, 6, 10, 13, 6, [70, 64, 65, 1]),
                                                (2, 130, 16, 320, 50, 4, [130, 77]), (17, 150, 12, 16, 9, 3, None),
                                                (3, 40, 5, 9, 7, 4, [40, 22, 7]), (2, 3, 4, 8, 5, 1, [3, 1]),
                                                (34, 20, 4, 8, 6, 2, None)])
def test_decoder_vs_oracle(B, T, He, Hd, V, L, lens):
    from asr_amd.Models.decoder import Decoder
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(num_hidden_nodes_encoder=He, num_hidden_nodes_decoder=Hd, num_classes=V)
    torch.manual_seed(T + Hd)
    ref = R.Decoder(hp).double()
    lengths = torch.tensor(lens) if lens is not None else torch.randint(1, T + 1, (B,))
    h = torch.randn(B, T, 2 * He)
    for b in range(B):
        h[b, lengths[b]:] = 0
    tgt = torch.randint(0, V, (B, L))
    hr = h.double().requires_grad_(True)
    yr = ref(hr, lengths, tgt)
    w = torch.randn(B, L, V).double()
    (yr * w).sum().backward()
    dec = Decoder(hp)
    dec.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    dec = dec.to(dev).train()
    hd = h.to(dev).requires_grad_(True)
    yd = dec(hd, lengths, tgt.to(dev))
    (yd * w.float().to(dev)).sum().backward()
    assert rel_err(yd.detach().cpu(), yr.detach()) < OUT_TOL
    assert rel_err(hd.grad.cpu(), hr.grad) < GRAD_TOL
    for (k, p), (_, q) in zip(dec.named_parameters(), ref.named_parameters()):
        # a parameter the reference never touched (L == 1) has no gradient there, zeros here
        qg = q.grad if q.grad is not None else torch.zeros_like(q)
        assert rel_err(p.grad.cpu(), qg) < GRAD_TOL, k


# ------------------------------------------------- golden vectors (reference) --
@pytest.mark.parametrize("name", CASES)
def test_golden_forward_backward(name):
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    z, hp = load_case(name)
    dev = _dev()
    m = _build(hp, params_of(z))
    x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"]).to(dev)
    tl = torch.from_numpy(z["text_lengths"]).to(dev)
    y = m(x, lengths, targets)
    loss = label_smoothing_loss(y, targets, tl, hp.T_norm, hp.B_norm)
    loss.backward()
    # anchor: the reference in float64 (its own float32 run differs from it by ~1e-6)
    assert rel_err(y.detach().cpu(), z["f64.youtput"]) < OUT_TOL
    assert abs(float(loss) - float(z["f64.loss"])) / abs(float(z["f64.loss"])) < LOSS_TOL
    assert rel_err(x.grad.cpu(), z["f64.dx"]) < GRAD_TOL
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), z[f"f64.grad.{k}"]) < GRAD_TOL, k
    # and against the reference's float32 outputs themselves
    assert rel_err(y.detach().cpu(), z["f32.youtput"]) < OUT_TOL
    assert abs(float(loss) - float(z["f32.loss"])) / abs(float(z["f32.loss"])) < LOSS_TOL


@pytest.mark.parametrize("name", CASES)
def test_golden_forward_backward_with_side_stream(name):
    """The same golden check with the weight gradients produced on the side stream
    (asr_amd/streams.py, what bench.py and BucketedGradReducer turn on), then a second backward
    without clearing .grad: with a gradient already in place the new one is added by autograd on
    the main stream, so the block must fall back to the main stream -- the sum is exactly 2x."""
    from asr_amd import streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    z, hp = load_case(name)
    dev = _dev()
    m = _build(hp, params_of(z))
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"]).to(dev)
    tl = torch.from_numpy(z["text_lengths"]).to(dev)
    if streams._FORCED_OFF:
        pytest.skip("ASR_SIDE_STREAM=0 forces the side stream off in this environment")
    streams.enable(True)
    try:
        assert streams.ENABLED
        x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
        loss = label_smoothing_loss(m(x, lengths, targets), targets, tl, hp.T_norm, hp.B_norm)
        loss.backward()
        assert rel_err(x.grad.cpu(), z["f64.dx"]) < GRAD_TOL
        first = {}
        for k, p in m.named_parameters():
            first[k] = p.grad.cpu().clone()          # .cpu() on the main stream: the join must have happened
            assert rel_err(first[k], z[f"f64.grad.{k}"]) < GRAD_TOL, k
        x2 = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
        loss2 = label_smoothing_loss(m(x2, lengths, targets), targets, tl, hp.T_norm, hp.B_norm)
        loss2.backward()                              # accumulates into the existing .grad
        for k, p in m.named_parameters():
            assert rel_err(p.grad.cpu(), 2.0 * first[k]) < 1e-5, k
    finally:
        streams.enable(False)


def test_side_stream_join_on_a_user_stream():
    """Forward/backward issued on a non-default stream with the side-stream overlap on: the
    end-of-backward join must make THAT stream wait for the weight gradients."""
    from asr_amd import streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    z, hp = load_case("mixed_h")
    dev = _dev()
    m = _build(hp, params_of(z))
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"]).to(dev)
    tl = torch.from_numpy(z["text_lengths"]).to(dev)
    x = torch.from_numpy(z["x"]).to(dev)
    user = torch.cuda.Stream()
    torch.cuda.synchronize()
    streams.enable(True)
    try:
        with torch.cuda.stream(user):
            for _ in range(3):
                for p in m.parameters():
                    p.grad = None
                loss = label_smoothing_loss(m(x, lengths, targets), targets, tl, hp.T_norm, hp.B_norm)
                loss.backward()
            grads = {k: p.grad.clone() for k, p in m.named_parameters()}   # reads on `user`, no device sync before
        user.synchronize()
        for k, g in grads.items():
            assert rel_err(g.cpu(), z[f"f64.grad.{k}"]) < GRAD_TOL, k
    finally:
        streams.enable(False)


@pytest.mark.parametrize("name", CASES)
def test_golden_greedy_tokens(name):
    z, hp = load_case(name)
    dev = _dev()
    m = _build(hp, params_of(z)).eval()
    x = torch.from_numpy(z["x"]).to(dev)
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(z["greedy.max_len"])
    hp.max_decoder_seq_len = max_len
    for utt in range(2):
        if f"greedy.utt{utt}" not in z.files:
            continue
        table = z[f"greedy.utt{utt}"]
        with torch.no_grad():
            hb = m.encoder(x[utt:utt + 1], lengths[utt:utt + 1])
            _, raw = m.decoder.decode_raw(hb, lengths[utt:utt + 1])
        assert len(raw) == max_len
        for eos in range(hp.num_classes):
            want = [int(v) for v in table[eos] if v >= 0]
            got = raw[:raw.index(eos) + 1] if eos in raw else []
            assert got == want, (utt, eos)
        # the public entry point with the reference's return convention
        hp.eos_id = int(raw[1])
        res = m.decode(x[utt:utt + 1], lengths[utt:utt + 1])
        assert res == [int(v) for v in table[hp.eos_id] if v >= 0]


# --------------------------------- production dims against the oracle, seeded --
def test_librispeech_dims_small_batch_vs_oracle():
    """hparams_template widths (H=320, 2 layers here, V=1000, stack 3 -> D=240) at a
    batch the CPU oracle finishes in seconds; exercises the 16-byte vector paths."""
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(num_encoder_layer=2, encoder_dropout=0.0)
    torch.manual_seed(1234)
    ref = R.AttModel(hp)
    B, T, L = 4, 48, 6
    lengths = torch.tensor([48, 40, 33, 20])
    tl = torch.tensor([6, 5, 3, 6])
    x = torch.randn(B, T, 240)
    tgt = torch.randint(0, 1000, (B, L))
    for b in range(B):
        x[b, lengths[b]:] = 0
        tgt[b, tl[b]:] = 0
    refd = R.AttModel(hp).double()
    refd.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = refd(xr, lengths, tgt)
    lr = R.label_smoothing_loss(yr, tgt, tl)
    lr.backward()
    m = _build(hp, ref.state_dict())
    xd = x.to(dev).requires_grad_(True)
    y = m(xd, lengths, tgt.to(dev))
    loss = label_smoothing_loss(y, tgt.to(dev), tl.to(dev))
    loss.backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < OUT_TOL
    assert abs(float(loss) - float(lr)) / abs(float(lr)) < LOSS_TOL
    assert rel_err(xd.grad.cpu(), xr.grad) < GRAD_TOL
    for (k, p), (_, q) in zip(m.named_parameters(), refd.named_parameters()):
        assert rel_err(p.grad.cpu(), q.grad) < GRAD_TOL, k


@pytest.mark.parametrize("cfg", ["config2_librispeech", "config4_csj"])
def test_full_size_properties(cfg):
    """BASELINE config 2 (B=32, T=500 stacked, L=100, V=1000, 5 layers) and config 4 (CSJ:
    40-dim fbank -> D=120, T_mel linspace(2400,1200) -> T=800, V=7000, L=150): too slow for
    the CPU oracle, so check size-independent properties: padded encoder rows are
    exactly zero, the loss is finite, gradients are finite, and the gradient of the loss
    along a random parameter direction matches a central finite difference."""
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    from bench import make_batch
    dev = _dev()
    torch.manual_seed(1234)
    if cfg == "config2_librispeech":
        hp = R.make_hp(encoder_dropout=0.0)
        m = AttModel(hp).to(dev).train()
        x, lengths, tgt, tl = make_batch(hp, dev, seed=1234)
    else:
        hp = R.make_hp(lmfb_dim=40, num_classes=7000, encoder_dropout=0.0)
        m = AttModel(hp).to(dev).train()
        x, lengths, tgt, tl = make_batch(hp, dev, seed=1234, t_hi=2400, t_lo=1200, L=150, l_lo=75)
        assert x.shape == (32, 800, 120)
    y = m(x, lengths, tgt)
    loss = label_smoothing_loss(y, tgt, tl)
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k
    with torch.no_grad():
        hb = m.encoder(x, lengths)
        for b in (0, 13, 31):
            n = int(lengths[b])
            if n < hb.shape[1]:
                assert float(hb[b, n:].abs().max()) == 0.0
        # directional derivative
        torch.manual_seed(7)
        dirs = {k: torch.randn_like(p) for k, p in m.named_parameters()}
        analytic = sum(float((p.grad.double() * dirs[k].double()).sum()) for k, p in m.named_parameters())
        eps = 2e-3
        vals = []
        for sgn in (+1, -1):
            for k, p in m.named_parameters():
                p.add_(dirs[k], alpha=sgn * eps)
            vals.append(float(label_smoothing_loss(m(x, lengths, tgt), tgt, tl)))
            for k, p in m.named_parameters():
                p.add_(dirs[k], alpha=-sgn * eps)
        fd = (vals[0] - vals[1]) / (2 * eps)
    assert abs(fd - analytic) / max(abs(analytic), 1e-6) < 2e-2, (fd, analytic)


def test_config1_speech_commands_shape_vs_oracle():
    """BASELINE config 1 (the reference's CPU-runnable case): B=8, 98 frames of 40-dim
    fbank, stack 3 -> T=32, D=120, template widths (H=320, 5 layers), V=37, L=3."""
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.utils.utils import frame_stacking
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=40, num_classes=37, encoder_dropout=0.0)
    torch.manual_seed(1234)
    ref = R.AttModel(hp)
    B, L = 8, 3
    mel = torch.randn(B, 98, 40)
    mel_len = torch.full((B,), 98, dtype=torch.int64)
    tgt = torch.randint(0, 37, (B, L))
    tl = torch.full((B,), L, dtype=torch.int64)
    x, lengths = frame_stacking(mel, mel_len, 3)
    x = x.contiguous()
    xr = x.clone().requires_grad_(True)
    lr = R.label_smoothing_loss(ref(xr, lengths, tgt), tgt, tl)
    lr.backward()
    m = _build(hp, ref.state_dict())
    xd = x.to(dev).requires_grad_(True)
    loss = label_smoothing_loss(m(xd, lengths, tgt.to(dev)), tgt.to(dev), tl.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(lr.detach())) / abs(float(lr.detach())) < LOSS_TOL
    assert rel_err(xd.grad.cpu(), xr.grad) < GRAD_TOL
    # a gradient that is zero up to cancellation (the bias inside the energy's tanh: ~5e-8 against 1e-2 for the
    # model's largest entry, with a float32 reference on the other side) is judged against 1e-3 of that largest entry
    floor = 1e-3 * max(float(q.grad.abs().max()) for q in ref.parameters())
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        err = float((p.grad.cpu() - q.grad).abs().max())
        assert err < GRAD_TOL * max(float(q.grad.abs().max()), floor), (k, err)


def test_dropout_between_layers():
    """encoder_dropout > 0 (template default 0.2): the mask is Bernoulli(1-p), scaled by
    1/(1-p), the same mask is applied in the backward, and eval() disables it."""
    import ctypes as C
    from asr_amd import _lib
    lib = _lib.load()
    dev = _dev()
    n, p = 1 << 20, 0.2
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    mask = torch.empty(n, device=dev, dtype=torch.uint8)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.asr_dropout_fwd(x.data_ptr(), y.data_ptr(), mask.data_ptr(), n, C.c_float(p), C.c_uint64(99), st))
    keep = mask.float().mean().item()
    assert abs(keep - (1 - p)) < 5e-3
    assert torch.equal(y, torch.where(mask.bool(), x * (1.0 / (1.0 - p)), torch.zeros_like(x)))
    dx = torch.empty_like(x)
    _lib.check(lib.asr_dropout_bwd(y.data_ptr(), mask.data_ptr(), dx.data_ptr(), n, C.c_float(p), st))
    assert torch.equal(dx, torch.where(mask.bool(), y * (1.0 / (1.0 - p)), torch.zeros_like(x)))
    # through the module: train() differs from eval(), eval() is deterministic
    from asr_amd.Models.encoder import Encoder
    from oracle import ref_model as R
    hp = R.make_hp(lmfb_dim=8, frame_stacking=1, num_hidden_nodes=16, num_encoder_layer=3, encoder_dropout=0.5)
    torch.manual_seed(5)
    enc = Encoder(hp).to(dev)
    xin = torch.randn(4, 20, 8, device=dev)
    lens = torch.tensor([20, 15, 9, 20])
    enc.eval()
    a, b = enc(xin, lens), enc(xin, lens)
    assert torch.equal(a, b)
    enc.train()
    c = enc(xin, lens)
    assert not torch.equal(a, c)
    xg = xin.clone().requires_grad_(True)
    enc(xg, lens).sum().backward()
    assert torch.isfinite(xg.grad).all()


# ------------------------------------------------ beam search --
@pytest.mark.parametrize("name", CASES)
def test_golden_beam_tokens(name):
    """Beam search on the device kernels (asr_attdec_step_att / _cell): token-identical to what
    the reference's decode() returned for every eos id (width 3 / log_softmax on utterances 0
    and 1, width 2 / softmax on utterance 0), and the surviving hypotheses of every step equal
    the oracle's (same token sequences in the same order, scores to 1e-4)."""
    from asr_amd.Models.beam import beam_decode
    from oracle import ref_model as R
    z, hp = load_case(name)
    dev = _dev()
    m = _build(hp, params_of(z)).eval()
    ref = R.AttModel(hp)
    ref.load_state_dict(params_of(z))
    ref.eval()
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(z["greedy.max_len"])
    for tag, width, score in (("beam3", 3, "log_softmax"), ("beam2sm", 2, "softmax")):
        for utt in range(2):
            if f"{tag}.utt{utt}" not in z.files:
                continue
            table = z[f"{tag}.utt{utt}"]
            with torch.no_grad():
                hb = m.encoder(x[utt:utt + 1].to(dev), lengths[utt:utt + 1])
                hr = ref.encoder(x[utt:utt + 1], lengths[utt:utt + 1])
            _, trace = beam_decode(m.decoder, hb, lengths[utt:utt + 1], width, max_len, -1, score)
            _, want_trace = ref.decoder.decode_beam(hr, lengths[utt:utt + 1], width, max_len, -1, score)
            assert len(trace) == len(want_trace) == max_len
            for got_step, want_step in zip(trace, want_trace):
                assert [s_ for s_, _ in got_step] == [s_ for s_, _ in want_step]
                for (_, a_), (_, b_) in zip(got_step, want_step):
                    assert abs(a_ - b_) < 1e-4 * max(1.0, abs(b_))
            for eos in range(hp.num_classes):
                want = [int(v) for v in table[eos] if v >= 0]
                got, _ = beam_decode(m.decoder, hb, lengths[utt:utt + 1], width, max_len, eos, score)
                assert got == want, (tag, utt, eos)
    # through the module interface (hp.beam_width, hp.score_func, hp.eos_id)
    hp.beam_width, hp.score_func = 3, "log_softmax"
    table = z["beam3.utt0"]
    for eos in range(min(hp.num_classes, 8)):
        hp.eos_id = eos
        got = m.decode(x[0:1].to(dev), lengths[0:1])
        assert got == [int(v) for v in table[eos] if v >= 0]


@pytest.mark.parametrize("name", V2_CASES)
def test_golden_beam_v2_tokens(name):
    """decode_v2 (Models/decoder.py:132-267) on the device kernels: token-identical to what the
    reference's decode_v2() returned for every eos id (widths 3 and 4 / log_softmax, 2 and 6 /
    softmax; [] where an empty result slot wins), and with an eos id that never occurs the live
    hypotheses of every step equal the oracle's (same sequences, same order, scores to 1e-4)."""
    from asr_amd.Models.beam import beam_decode_v2
    from oracle import ref_model as R
    z, hp = load_case(name)
    v2 = load_v2(name)
    dev = _dev()
    m = _build(hp, params_of(z)).eval()
    ref = R.AttModel(hp)
    ref.load_state_dict(params_of(z))
    ref.eval()
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(v2["max_len"])
    for tag, width, score in V2_RUNS:
        for utt in range(2):
            if f"{tag}.utt{utt}" not in v2.files:
                continue
            table = v2[f"{tag}.utt{utt}"]
            with torch.no_grad():
                hb = m.encoder(x[utt:utt + 1].to(dev), lengths[utt:utt + 1])
                hr = ref.encoder(x[utt:utt + 1], lengths[utt:utt + 1])
            _, trace = beam_decode_v2(m.decoder, hb, lengths[utt:utt + 1], width, max_len, -1, score)
            _, want_trace = ref.decoder.decode_beam_v2(hr, lengths[utt:utt + 1], width, max_len, -1, score)
            assert len(trace) == len(want_trace) == max_len
            for (got_live, _), (want_live, _) in zip(trace, want_trace):
                assert [s_ for s_, _ in got_live] == [s_ for s_, _ in want_live]
                for (_, a_), (_, b_) in zip(got_live, want_live):
                    assert abs(a_ - b_) < 1e-4 * max(1.0, abs(b_))
            for eos in range(hp.num_classes):
                want = [int(v) for v in table[eos] if v >= 0]
                got, _ = beam_decode_v2(m.decoder, hb, lengths[utt:utt + 1], width, max_len, eos, score)
                assert got == want, (tag, utt, eos)
    # through the module interface (hp.beam_width, hp.score_func, hp.eos_id, hp.max_decoder_seq_len)
    hp.beam_width, hp.score_func, hp.max_decoder_seq_len = 3, "log_softmax", max_len
    table = v2["v2beam3.utt0"]
    for eos in range(hp.num_classes):
        hp.eos_id = eos
        assert m.decode_v2(x[0:1].to(dev), lengths[0:1]) == [int(v) for v in table[eos] if v >= 0]


def test_decoder_loop_pauses_do_not_change_results():
    """asr_attdec_pace: the pauses in front of the loop kernels' waits are timing only -- with every pause at 0 and with
    long ones (3 to 63 units) the outputs and the data gradient are bit-identical to the defaults' (the weight gradients
    agree to the run-to-run order of their atomic split-K sums), and no wait times out."""
    from asr_amd import _lib
    from asr_amd.Models.decoder import Decoder
    from asr_amd.Models.encoder import persistent_error_flags
    from oracle import ref_model as R
    lib = _lib.load()
    dev = _dev()
    B, T, He, Hd, V, L = 32, 500, 320, 320, 100, 6
    assert lib.asr_attdec_plan(B, T, 2 * He, Hd, L) == 3
    hp = R.make_hp(num_hidden_nodes_encoder=He, num_hidden_nodes_decoder=Hd, num_classes=V)
    torch.manual_seed(5)
    dec = Decoder(hp).to(dev).train()
    lengths = torch.linspace(T, int(0.6 * T), B).long()
    h = torch.randn(B, T, 2 * He)
    for b in range(B):
        h[b, lengths[b]:] = 0
    tgt, w = torch.randint(0, V, (B, L)).to(dev), torch.randn(B, L, V).to(dev)
    defaults = [lib.asr_attdec_pace(i, -1) for i in range(16)]
    assert defaults[2] > 0 and lib.asr_attdec_pace(16, 0) == -1
    res = []
    try:
        for setting in (defaults, [0] * 16, [3 + 4 * i for i in range(16)]):
            for i, v in enumerate(setting):
                lib.asr_attdec_pace(i, v)
            dec.zero_grad(set_to_none=True)
            hd = h.to(dev).requires_grad_(True)
            y = dec(hd, lengths, tgt)
            (y * w).sum().backward()
            res.append([y.detach().clone(), hd.grad.clone()] + [p.grad.clone() for p in dec.parameters()])
    finally:
        for i, v in enumerate(defaults):
            lib.asr_attdec_pace(i, v)
    assert not persistent_error_flags()
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])      # youtput, d hbatch
        for a_, b_ in zip(res[0][2:], other[2:]):       # weight gradients: split-K sums with float atomics, order varies
            assert rel_err(b_.cpu(), a_.cpu()) < 1e-5


def test_beam_v2_errors():
    """The reference raises RuntimeError for unequal encoder/decoder widths (decoder.py:219,241)
    and for widths its topk calls reject; an LM together with softmax scoring is refused."""
    from asr_amd.Models.beam import beam_decode_v2
    z, hp = load_case("mixed_h")
    dev = _dev()
    m = _build(hp, params_of(z)).eval()
    x = torch.from_numpy(z["x"])[:1].to(dev)
    lengths = torch.from_numpy(z["lengths"])[:1]
    with torch.no_grad():
        hb = m.encoder(x, lengths)
    with pytest.raises(RuntimeError):
        beam_decode_v2(m.decoder, hb, lengths, 3, 5, 1)
    z, hp = load_case("tiny_a")
    m = _build(hp, params_of(z)).eval()
    x = torch.from_numpy(z["x"])[:1].to(dev)
    lengths = torch.from_numpy(z["lengths"])[:1]
    with torch.no_grad():
        hb = m.encoder(x, lengths)
    for width in (1, hp.num_classes + 1):
        with pytest.raises(RuntimeError):
            beam_decode_v2(m.decoder, hb, lengths, width, 5, 1)
    hp.score_func, hp.lm_weight, hp.beam_width = "softmax", 0.5, 3      # an LM with softmax scoring is broken in the reference: refused
    with pytest.raises(ValueError):
        m.decode_v2(x, lengths, model_lm=lambda t: torch.zeros(t.shape[0], t.shape[1], hp.num_classes, device=t.device))


@pytest.mark.parametrize("name", LM_CASES)
def test_golden_shallow_fusion_tokens(name):
    """decode / decode_v2 with a language model (Models/decoder.py:93-96, :180-183) on the device kernels:
    token-identical to what the REFERENCE returned with the same LM (tests/golden/lm: widths 1 and 3 of decode, width
    3 of decode_v2, lm_weight 0.7) for every eos id, through the module interface; and with an eos id that never
    occurs the survivors of every step equal the oracle's (same sequences, same order, scores to 1e-4)."""
    from asr_amd.Models.beam import beam_decode, beam_decode_v2
    from oracle import ref_model as R
    z, hp = load_case(name)
    g, lm, lm_weight = load_lm(name)
    dev = _dev()
    m = _build(hp, params_of(z)).eval()
    ref = R.AttModel(hp)
    ref.load_state_dict(params_of(z))
    ref.eval()
    lm_dev = copy.deepcopy(lm).to(dev)
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(g["max_len"])
    hp.max_decoder_seq_len, hp.lm_weight, hp.score_func = max_len, lm_weight, "log_softmax"
    changed = 0
    for tag, fn, width in LM_RUNS:
        hp.beam_width = width
        for utt in range(2):
            xu, lu = x[utt:utt + 1], lengths[utt:utt + 1]
            with torch.no_grad():
                hb, hr = m.encoder(xu.to(dev), lu), ref.encoder(xu, lu)
            if fn == "decode":
                _, trace = beam_decode(m.decoder, hb, lu, width, max_len, -1, "log_softmax", lm_dev, lm_weight)
                _, want_trace = ref.decoder.decode_beam(hr, lu, width, max_len, -1, "log_softmax", lm, lm_weight)
            else:
                _, trace = beam_decode_v2(m.decoder, hb, lu, width, max_len, -1, "log_softmax", lm_dev, lm_weight)
                _, want_trace = ref.decoder.decode_beam_v2(hr, lu, width, max_len, -1, "log_softmax", lm, lm_weight)
                trace, want_trace = [t[0] for t in trace], [t[0] for t in want_trace]
            assert len(trace) == len(want_trace) == max_len
            for got_step, want_step in zip(trace, want_trace):
                assert [s_ for s_, _ in got_step] == [s_ for s_, _ in want_step]
                for (_, a_), (_, b_) in zip(got_step, want_step):
                    assert abs(a_ - b_) < 1e-4 * max(1.0, abs(b_))
            table = g[f"{tag}.utt{utt}"]
            for eos in range(hp.num_classes):
                hp.eos_id = eos
                got = getattr(m, fn)(xu.to(dev), lu, lm_dev)
                assert got == [int(v) for v in table[eos] if v >= 0], (tag, utt, eos)
                changed += got != getattr(m, fn)(xu.to(dev), lu)
    assert changed > 0


# ------------------------------------------------ optimizer step --
@pytest.mark.parametrize("max_norm", [5.0, 0.05, None])
def test_clip_adam_vs_torch(max_norm):
    """asr_grad_sqnorm + asr_clip_adam_step against torch.nn.utils.clip_grad_norm_ +
    torch.optim.Adam(weight_decay=1e-5) (train.py:60-63,146) in float64 on the CPU: four steps on
    tensors of awkward sizes (1 element, odd, > one block's stride); max_norm 5.0 does not clip
    these gradients, 0.05 does.  Tolerance 2e-6 relative (fp32 update of O(1) parameters)."""
    from asr_amd.optim import ClipAdam
    dev = _dev()
    torch.manual_seed(3)
    shapes = [(1,), (7,), (33, 5), (1280, 640), (1000, 1281), (3, 4, 5)]
    ps = [torch.randn(*sh) for sh in shapes]
    ref = [torch.nn.Parameter(p.double().clone()) for p in ps]
    got = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    ropt = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-5)
    gopt = ClipAdam(got, lr=1e-3, weight_decay=1e-5)
    for step in range(4):
        gs = [torch.randn(*sh) * (0.1 + step) for sh in shapes]
        for p, q, g in zip(ref, got, gs):
            p.grad = g.double().clone()
            q.grad = g.clone().to(dev)
        if max_norm is not None:
            tn = torch.nn.utils.clip_grad_norm_(ref, max_norm)
        ropt.step()
        out = gopt.step(max_norm=max_norm)
        if max_norm is not None:
            assert abs(float(out) - float(tn)) / float(tn) < 1e-6
            for p, q in zip(ref, got):                      # the clipped gradient is written back
                assert rel_err(q.grad.cpu(), p.grad) < 2e-6
        for p, q in zip(ref, got):
            assert rel_err(q.detach().cpu(), p.detach()) < 2e-6
            assert rel_err(gopt.state[q]["exp_avg"].cpu(), ropt.state[p]["exp_avg"]) < 2e-6
            assert rel_err(gopt.state[q]["exp_avg_sq"].cpu(), ropt.state[p]["exp_avg_sq"]) < 2e-6
    # state_dict layout is torch.optim.Adam's: it loads into a stock Adam on the same parameters
    stock = torch.optim.Adam(got, lr=1e-3, weight_decay=1e-5)
    stock.load_state_dict(gopt.state_dict())
    assert int(stock.state[got[0]]["step"]) == 4


def test_clip_adam_rejects_cpu_parameters():
    from asr_amd.optim import ClipAdam
    p = torch.nn.Parameter(torch.randn(4))
    p.grad = torch.randn(4)
    with pytest.raises(RuntimeError):
        ClipAdam([p]).step()


# ------------------------------------------------- failure surfacing / guards --
def test_timed_out_wait_poisons_the_loss():
    """A persistent recurrence kernel that gives up on an in-launch wait sets the sticky error word of
    its workspace; label_smoothing_loss must then return NaN (the reference's loop exits on a NaN
    loss, train.py:66-68) -- simulated by setting the word by hand, then cleared again."""
    from asr_amd import persist
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.encoder import Encoder
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=6, frame_stacking=1, num_hidden_nodes=16, num_encoder_layer=1, encoder_dropout=0.0)
    enc = Encoder(hp).to(dev)
    xin, lens = torch.randn(3, 9, 6, device=dev), torch.tensor([9, 5, 2])
    enc(xin, lens)      # makes sure a workspace exists
    assert persist._WS and not persist.error_flags()
    z = torch.randn(2, 3, 12, device=dev, requires_grad=True)
    tgt = torch.randint(0, 12, (2, 3), device=dev)
    tl = torch.tensor([3, 2], device=dev)
    clean = label_smoothing_loss(z, tgt, tl)
    assert bool(torch.isfinite(clean))
    ws = persist._WS[(dev.index, 3, 16, torch.cuda.current_stream(dev).cuda_stream)]      # this encoder's: B = 3, H = 16
    try:
        ws.view(torch.int32)[persist.ERR_WORD] = 1
        assert persist.error_flags()
        # the loss looks at the workspaces used since the last loss: a launch that used this one (the only way the
        # word gets set) puts it on that list; the launch's own set-up never clears the sticky word
        enc(xin, lens)
        assert persist.error_flags()
        assert bool(torch.isnan(label_smoothing_loss(z, tgt, tl)))
        # ... and a backward launch folds it into its layer's bias gradients at once
        enc.zero_grad(set_to_none=True)
        enc(xin, lens).sum().backward()
        assert bool(torch.isnan(enc.bi_lstm.bias_ih_l0.grad[0])) and bool(torch.isnan(enc.bi_lstm.bias_hh_l0.grad[0]))
        # one NaN entry makes the global norm NaN: the fused clip + Adam then writes NaN everywhere, as stock
        # clip_grad_norm_ + Adam do -- the job cannot train on past the failure
        from asr_amd.optim import ClipAdam
        import copy
        enc2 = copy.deepcopy(enc)
        for p2, p1 in zip(enc2.parameters(), enc.parameters()):
            p2.grad = p1.grad.clone()
        ClipAdam(enc2.parameters(), lr=1e-3).step(max_norm=5.0)
        assert all(bool(torch.isnan(p2).all()) for p2 in enc2.parameters())
    finally:
        ws.view(torch.int32)[persist.ERR_WORD] = 0
    enc(xin, lens)
    assert float(label_smoothing_loss(z, tgt, tl)) == float(clean)
    enc.zero_grad(set_to_none=True)
    enc(xin, lens).sum().backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in enc.parameters())


@pytest.mark.parametrize("n_wg,threads,lds,usec", [(32, 256, 16384, 250), (64, 512, 32768, 400), (256, 256, 8192, 150)])
def test_step_under_competing_kernels(n_wg, threads, lds, usec):
    """One training step at the config-2 size while ANOTHER stream keeps launching kernels that sit on compute units
    and do nothing (asr_debug_occupy: a stand-in for RCCL's channel workgroups, which stay resident for the length of
    a collective): the persistent kernels need all their workgroups resident at once, so while such a kernel is in
    the way some of them start late and the others wait for them -- in-launch waits are bounded by wall time (4 s),
    far beyond any collective, so nothing may time out: no error word, finite loss, and the gradients of the quiet
    run (split-K float atomics reorder sums: 2e-5).  The competing launches cover the whole step and end by
    themselves (each spins on the clock for `usec`)."""
    import ctypes as C
    from asr_amd import _lib, persist, streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    lib = _lib.load()
    dev = _dev()
    hp = R.make_hp(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5, num_classes=1000,
                   encoder_dropout=0.0)
    torch.manual_seed(77)
    m = AttModel(hp).to(dev).train()
    B, T, L = 32, 500, 24
    lengths = torch.linspace(T, 300, B).long()
    x = torch.randn(B, T, 240)
    for b in range(B):
        x[b, lengths[b]:] = 0
    x = x.to(dev)
    tgt = torch.randint(0, 1000, (B, L)).to(dev)
    tl = torch.linspace(L, L // 2, B).long().to(dev)

    def step():
        m.zero_grad(set_to_none=True)
        loss = label_smoothing_loss(m(x, lengths, tgt), tgt, tl, hp.T_norm, hp.B_norm)
        loss.backward()
        return loss

    was = streams.ENABLED
    streams.enable(True)          # as the data-parallel job runs: weight gradients on the side stream
    try:
        step()
        quiet = step()
        torch.cuda.synchronize()
        want = {k: p.grad.clone() for k, p in m.named_parameters()}
        other = torch.cuda.Stream(device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_launch = max(8, int(60000 / usec))                      # ~60 ms of competing kernels, back to back
        for _ in range(n_launch // 8):                            # a head start: they are resident when the step begins
            _lib.check(lib.asr_debug_occupy(n_wg, threads, lds, usec, C.c_void_p(other.cuda_stream)))
        e0.record()
        for i in range(n_launch):
            _lib.check(lib.asr_debug_occupy(n_wg, threads, lds, usec, C.c_void_p(other.cuda_stream)))
            if i == n_launch // 8:
                loss = step()
                e1.record()
        torch.cuda.synchronize()
    finally:
        streams.enable(was)
    assert not persist.error_flags()
    assert bool(torch.isfinite(loss)) and abs(float(loss) - float(quiet)) < 1e-5 * abs(float(quiet))
    floor = 1e-3 * max(float(v.abs().max()) for v in want.values())
    for k, p in m.named_parameters():
        assert bool(torch.isfinite(p.grad).all()), k
        err = float((p.grad - want[k]).abs().max())
        assert err < 2e-5 * max(float(want[k].abs().max()), floor), (k, err)
    assert e0.elapsed_time(e1) < 2000.0       # the step only waited; it did not sit out a 4 s wait bound


def test_second_backward_raises_clearly():
    """The BLSTM backward consumes its saved gates in place and the decoder releases its saved
    tensors: a second pass over a retained graph must raise, not return wrong gradients."""
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=6, frame_stacking=1, num_hidden_nodes=8, num_encoder_layer=1, num_classes=9,
                   encoder_dropout=0.0)
    torch.manual_seed(2)
    m = AttModel(hp).to(dev).train()
    x = torch.randn(2, 7, 6, device=dev)
    y = m(x, torch.tensor([7, 4]), torch.randint(0, 9, (2, 3), device=dev))
    y.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        y.sum().backward()
    h = m.encoder(x, torch.tensor([7, 4]))
    h.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second time"):
        h.sum().backward()


def test_blstm_data_parallel_rehearsal_group_of_one_then_side_stream():
    """The N-GPU code path of asr_amd.parallel on the BLSTM branch in an RCCL group of one (hooks,
    bucket packing from the side stream, all-reduce, mean, scatter back): gradients equal a plain
    backward.  The reducer is closed before the group is destroyed, and a side-stream backward
    must still run afterwards in the same process (a training job that builds a second group for
    evaluation or resume) -- which is why this test is NOT collected last."""
    import os
    import socket
    import torch.distributed as dist
    from asr_amd import streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.parallel import BucketedGradReducer
    from oracle import ref_model as R
    if streams._FORCED_OFF:
        pytest.skip("ASR_SIDE_STREAM=0 forces the side stream off in this environment")
    dev = _dev()
    hp = R.make_hp(lmfb_dim=12, frame_stacking=1, num_hidden_nodes=32, num_encoder_layer=3, num_classes=21,
                   encoder_dropout=0.0)
    torch.manual_seed(5)
    m = AttModel(hp).to(dev).train()
    B, T, L = 4, 70, 5
    x = torch.randn(B, T, 12, device=dev)
    lengths = torch.tensor([70, 66, 41, 12])
    targets = torch.randint(0, 21, (B, L), device=dev)
    tl = torch.tensor([5, 4, 3, 5], device=dev)

    def grads():
        m.zero_grad(set_to_none=True)
        label_smoothing_loss(m(x, lengths, targets), targets, tl).backward()

    was = streams.ENABLED
    grads()
    torch.cuda.synchronize()
    want = {k: p.grad.clone() for k, p in m.named_parameters()}
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        with BucketedGradReducer(m, force=True) as red:
            assert red.active and len(red.buckets) == 1 + 3 and streams.ENABLED
            red.broadcast_parameters()
            for _ in range(2):
                grads()
                red.finish()
            torch.cuda.synchronize()
            assert all(w is None for w in red.work)
            for k, p in m.named_parameters():
                assert rel_err(p.grad.cpu(), want[k].cpu()) < 1e-6, k
        assert streams.ENABLED == was and all(st.query() for st in streams._streams.values())
    finally:
        dist.destroy_process_group()
    # afterwards: the side stream (a fresh one) must still work in this process
    streams.enable(True)
    try:
        grads()
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            assert rel_err(p.grad.cpu(), want[k].cpu()) < 1e-6, k
    finally:
        streams.enable(was)


# ------------------------------------------------------- GEMM arithmetic modes --
@pytest.mark.parametrize("mode,loss_tol,out_tol,grad_tol", [("f32", 1e-4, 1e-4, 1e-3), ("bf16x3", 1e-4, 1e-4, 1e-3),
                                                            ("bf16", 1e-3, 2e-2, 5e-2)])
def test_attmodel_gemm_precision_modes(mode, loss_tol, out_tol, grad_tol):
    """The whole AttModel step at production widths (5 x BLSTM 320, attention 640, V = 1000) against the float64
    oracle in each arithmetic mode of the hoisted matrix products: exact fp32 and the bf16x3 split (the default)
    are held to the north-star bar (loss / youtput 1e-4, gradients 1e-3); plain bf16 products -- the opt-in
    reduced-precision path -- to its own, stated, looser one (loss 1e-3, youtput 2e-2, gradients 5e-2).
    Gradients: max |err| over the tensor / max(|tensor|max, 1e-3 of the model's largest gradient entry): at random
    initialisation the attention's own gradients are ~1e-8 of the largest, where fp32 itself has no digits left."""
    from asr_amd import ops
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5, num_classes=1000,
                   encoder_dropout=0.0)
    torch.manual_seed(31)
    ref = R.AttModel(hp).double()
    B, T, L = 6, 48, 9
    lengths = torch.tensor([48, 48, 40, 33, 20, 7])
    x = torch.randn(B, T, 240)
    for b in range(B):
        x[b, lengths[b]:] = 0
    tgt = torch.randint(0, 1000, (B, L))
    tl = torch.tensor([9, 7, 9, 4, 6, 2])
    yr = ref(x.double(), lengths, tgt)
    lr = R.label_smoothing_loss(yr, tgt, tl, hp.T_norm, hp.B_norm)
    lr.backward()
    m = AttModel(hp)
    m.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    m = m.to(dev).train()
    with ops.gemm_precision(mode):
        y = m(x.to(dev), lengths, tgt.to(dev))
        loss = label_smoothing_loss(y, tgt.to(dev), tl.to(dev), hp.T_norm, hp.B_norm)
        loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(lr.detach())) < loss_tol * abs(float(lr.detach()))
    assert rel_err(y.detach().cpu(), yr.detach()) < out_tol
    want = dict(ref.named_parameters())
    floor = 1e-3 * max(float(q.grad.abs().max()) for q in want.values())
    for k, p in m.named_parameters():
        err = float((p.grad.cpu().double() - want[k].grad).abs().max())
        assert err < grad_tol * max(float(want[k].grad.abs().max()), floor), (k, err)


@pytest.mark.parametrize("B,T,He,Hd,V,L,lens", [(32, 500, 320, 320, 1000, 12, "ragged"), (5, 130, 16, 8, 11, 6, [130, 64, 65, 1, 100]),
                                                (32, 374, 128, 320, 1000, 7, "ragged"),      # config-5 decoder: E = 256
                                                (32, 448, 320, 320, 1000, 5, "full"),        # no empty tile: X jobs on loaded tiles
                                                (16, 512, 320, 320, 1000, 4, "full"),        # one batch tile, T a multiple of 64
                                                (16, 800, 320, 320, 1000, 4, "full"),        # 13 tiles: 22 frames per thread in the context sum
                                                (32, 800, 320, 320, 7000, 6, "ragged"),      # config-4 (CSJ) decoder: two parts of 16 utterances
                                                (32, 800, 320, 320, 7000, 3, "ragged+side"),  # ... its weight gradients summed on the side stream
                                                (18, 70, 8, 16, 9, 4, None), (3, 17, 4, 8, 7, 5, [17, 9, 4])])
def test_decoder_persistent_loop_vs_per_step_launches(B, T, He, Hd, V, L, lens):
    """The one-launch decoder loop (in-launch hand-offs between workgroups) against the four-launches-per-step
    loop on the same inputs: outputs and every gradient (the backward consumes what the forward saved: S, C,
    alpha, sproj, g, gates, Fc).  The two differ only in the order of the context sums.  Every shape here must TAKE
    the one-launch form (asr_attdec_plan), the config-4 shape as two parts of 16 utterances (416 workgroups do not
    fit 256 compute units; Decoder splits the batch, asr_amd/Models/decoder.py::batch_parts)."""
    from asr_amd import _lib, streams
    from asr_amd.Models.decoder import Decoder, batch_parts
    from oracle import ref_model as R
    lib = _lib.load()
    dev = _dev()
    side = isinstance(lens, str) and lens.endswith("+side")
    if side:
        lens = lens[:-5]
    parts = batch_parts(lib, B, T, 2 * He, Hd, L)
    assert all(lib.asr_attdec_plan(b1 - b0, T, 2 * He, Hd, L) == 3 for b0, b1 in parts), parts
    assert len(parts) == (2 if (B, T) == (32, 800) else 1), parts
    assert sum(b1 - b0 for b0, b1 in parts) == B and parts[0][0] == 0 and parts[-1][1] == B
    hp = R.make_hp(num_hidden_nodes_encoder=He, num_hidden_nodes_decoder=Hd, num_classes=V)
    torch.manual_seed(T + Hd)
    dec = Decoder(hp).to(dev).train()
    if lens == "full":
        lengths = torch.full((B,), T, dtype=torch.int64)
    elif lens == "ragged":
        lengths = torch.linspace(T, int(0.6 * T), B).long()
    else:
        lengths = torch.tensor(lens) if lens is not None else torch.randint(1, T + 1, (B,))
    h = torch.randn(B, T, 2 * He)
    for b in range(B):
        h[b, lengths[b]:] = 0
    tgt = torch.randint(0, V, (B, L)).to(dev)
    w = torch.randn(B, L, V).to(dev)
    res = []
    was, side_was = lib.asr_attdec_persist(-1), streams.ENABLED
    try:
        for on in (0, 1):
            lib.asr_attdec_persist(on)
            streams.enable(side and on == 1)
            dec.zero_grad(set_to_none=True)
            hd = h.to(dev).requires_grad_(True)
            y = dec(hd, lengths, tgt)
            (y * w).sum().backward()
            torch.cuda.synchronize()
            res.append((y.detach().clone(), hd.grad.clone(), {k: p.grad.clone() for k, p in dec.named_parameters()}))
    finally:
        lib.asr_attdec_persist(was)
        streams.enable(side_was)
    (y0, dh0, g0), (y1, dh1, g1) = res
    assert bool(torch.isfinite(y1).all())
    assert rel_err(y1.cpu(), y0.cpu()) < 1e-5
    assert rel_err(dh1.cpu(), dh0.cpu()) < 2e-5
    floor = 1e-3 * max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        err = float((g1[k] - g0[k]).abs().max())
        assert err < 2e-5 * max(float(g0[k].abs().max()), floor), (k, err)
