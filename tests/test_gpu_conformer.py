"""GPU parity of the Conformer encoder through the C ABI: asr_amd.Models.Transformer_encoder
against the golden vectors produced by the reference's Models/Transformer_encoder.py and, at production
widths, against the CPU restatement.  Bar: outputs 1e-4, gradients 1e-3 (max-abs error relative to the
max-abs value; gradients that are zero in exact arithmetic are judged against 5 % of the largest one)."""
import numpy as np
import pytest
import torch

from tests._util import rel_err
from tests.test_conformer_oracle import CONF_CASES, conf_state, grad_err, grad_floor, load_conf

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _build(meta, state, dropout=0.0):
    from asr_amd.Models.Transformer_encoder import TransformerEncoder
    m = TransformerEncoder(meta["idim"], meta["d_model"], meta["N_e"], meta["heads"], dropout)
    m.load_state_dict(state)
    return m.to(_dev())


@pytest.mark.parametrize("name", CONF_CASES)
def test_golden_conformer(name):
    z, meta = load_conf(name)
    dev = _dev()
    m = _build(meta, conf_state(z))
    assert sorted(m.state_dict().keys()) == sorted(k[len("param."):] for k in z.files if k.startswith("param."))
    src = torch.from_numpy(z["src"]).to(dev)
    src_pos = torch.from_numpy(z["src_pos"]).to(dev)
    m.eval()
    with torch.no_grad():
        assert rel_err(m(src, src_pos).cpu(), z["f32.eval_out"]) < 1e-4
    m.train()
    y = m(src, src_pos)
    assert rel_err(y.detach().cpu(), z["f32.train_out"]) < 1e-4
    (y * torch.from_numpy(z["R"]).to(dev)).sum().backward()
    floor = grad_floor(z, "f32")
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert grad_err(p.grad.cpu(), z[f"f32.grad.{k}"], floor) < 1e-3, k
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v.cpu(), z[f"f32.after.{k}"]) < 1e-5, k
        if "num_batches" in k:
            assert int(v) == int(z[f"f32.after.{k}"])


def test_production_widths_vs_oracle():
    """d_model 256, 4 heads, 80-dim features (AttModel.py:17) with 2 layers, ragged batch: output, all
    gradients and BatchNorm statistics against the CPU restatement."""
    from oracle import ref_conformer as RC
    dev = _dev()
    torch.manual_seed(7)
    ref = RC.TransformerEncoder(80, 256, 2, 4, 0.0)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if "norm" in k:
                p.add_(0.1 * torch.randn_like(p))
    B, T, lens = 3, 423, [423, 300, 171]
    src = torch.randn(B, T, 80)
    src_pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        src[b, n:] = 0
        src_pos[b, :n] = torch.arange(1, n + 1)
    m = _build(dict(idim=80, d_model=256, N_e=2, heads=4), ref.state_dict())
    ref.train()
    m.train()
    yr = ref(src, src_pos)
    R = torch.randn_like(yr)
    (yr * R).sum().backward()
    y = m(src.to(dev), src_pos.to(dev))
    assert y.shape == yr.shape == (B, 105, 256)
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    (y * R.to(dev)).sum().backward()
    want = dict(ref.named_parameters())
    floor = 5e-2 * max(float(p.grad.abs().max()) for p in want.values())
    for k, p in m.named_parameters():
        assert grad_err(p.grad.cpu(), want[k].grad, floor) < 1e-3, k
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v.cpu(), ref.state_dict()[k]) < 1e-5, k


def test_dropout_modes_and_errors():
    """Training with dropout 0.1 differs from eval and stays finite with finite gradients; eval is
    deterministic; CPU input raises."""
    z, meta = load_conf("conf_tiny")
    dev = _dev()
    m = _build(meta, conf_state(z), dropout=0.1)
    src = torch.from_numpy(z["src"]).to(dev)
    src_pos = torch.from_numpy(z["src_pos"]).to(dev)
    m.eval()
    with torch.no_grad():
        a, b = m(src, src_pos), m(src, src_pos)
    assert torch.equal(a, b)
    m.train()
    y = m(src, src_pos)
    assert torch.isfinite(y).all() and not torch.equal(y.detach(), a)
    y.sum().backward()
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    with pytest.raises(RuntimeError):
        m(src.cpu(), src_pos.cpu())


def test_attmodel_conformer_branch_vs_oracle():
    """AttModel(encoder_type='Conformer') (Models/AttModel.py:16-17,23-27: 16 layers, d_model 256, 4 heads,
    the decoder fed the un-subsampled lengths) + label-smoothed CE: loss and every gradient against the CPU
    restatements (Conformer encoder + attention decoder), dropout off so the training-mode pass is deterministic."""
    import torch.nn as nn
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_conformer as RC
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=20, frame_stacking=1, num_hidden_nodes_encoder=256, num_hidden_nodes_decoder=24,
                   num_classes=31, encoder_type="Conformer", encoder_dropout=0.0)
    torch.manual_seed(3)
    m = AttModel(hp)

    class RefAtt(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = RC.TransformerEncoder(hp.lmfb_dim, 256, 16, 4, 0.0)
            self.decoder = R.Decoder(hp)

        def forward(self, x, lengths, targets, src_pos):
            return self.decoder(self.encoder(x, src_pos), lengths, targets)

    ref = RefAtt()
    ref.load_state_dict(m.state_dict())
    for mod in m.modules():
        if hasattr(mod, "p"):
            mod.p = 0.0
    m = m.to(dev).train()
    ref.train()
    B, T, L, lens, tls = 2, 90, 4, [90, 61], [4, 2]
    x = torch.randn(B, T, 20)
    src_pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        x[b, n:] = 0
        src_pos[b, :n] = torch.arange(1, n + 1)
    targets = torch.randint(0, 31, (B, L))
    lengths, tl = torch.tensor(lens), torch.tensor(tls)
    yr = ref(x, lengths, targets, src_pos)
    lr = R.label_smoothing_loss(yr, targets, tl)
    lr.backward()
    y = m(x.to(dev), lengths, targets.to(dev), src_pos.to(dev))
    loss = label_smoothing_loss(y, targets.to(dev), tl.to(dev))
    loss.backward()
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    assert abs(float(loss.detach()) - float(lr.detach())) / abs(float(lr.detach())) < 1e-4
    want = dict(ref.named_parameters())
    floor = 5e-2 * max(float(p.grad.abs().max()) for p in want.values())
    for k, p in m.named_parameters():
        assert grad_err(p.grad.cpu(), want[k].grad, floor) < 1e-3, k
    # greedy and beam decode through the same branch (Models/AttModel.py:31-39): eval mode, batch 1, src_pos passed on
    m.eval()
    ref.eval()
    hp.max_decoder_seq_len = 6
    with torch.no_grad():
        hr = ref.encoder(x[:1], src_pos[:1])
    _, raw = ref.decoder.decode_greedy(hr, lengths[:1], 6, eos_id=-1)
    hp.beam_width, hp.eos_id = 1, raw[2]
    assert m.decode(x[:1].to(dev), lengths[:1], None, src_pos[:1].to(dev)) == raw[:raw.index(raw[2]) + 1]
    hp.beam_width = 3
    want, _ = ref.decoder.decode_beam(hr, lengths[:1], 3, 6, hp.eos_id)
    assert m.decode(x[:1].to(dev), lengths[:1], None, src_pos[:1].to(dev)) == want


@pytest.mark.parametrize("rows,d", [(37, 100), (50, 300), (9, 700), (5, 1024), (130, 64)])
def test_layernorm_kernels_all_widths(rows, d):
    """asr_layernorm_fwd / _bwd at widths that exercise every register-count instantiation (d <= 64 ... 1024),
    against torch.nn.functional.layer_norm in float64; the backward adds the residual-path gradient."""
    from asr_amd.Models.Transformer_encoder import _ln_bwd, _ln_bwd_sums, _ln_fwd
    dev = _dev()
    torch.manual_seed(d)
    x, g, b = torch.randn(rows, d), torch.randn(d), torch.randn(d)
    dy, res = torch.randn(rows, d), torch.randn(rows, d)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
    yr.backward(dy.double())
    y, mean, rstd = _ln_fwd(x.to(dev), g.to(dev), b.to(dev))
    assert rel_err(y.cpu(), yr.detach()) < 1e-5
    dx, part = _ln_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, res.to(dev))
    dg, db = _ln_bwd_sums(part)
    assert rel_err(dx.cpu(), xr.grad + res.double()) < 1e-5
    assert rel_err(dg.cpu(), gr.grad) < 1e-5
    assert rel_err(db.cpu(), br.grad) < 1e-5


def test_dropout_backward_matches_finite_differences():
    """With dropout 0.1 the masks come from seeds drawn from torch's CPU generator, so re-seeding reproduces them:
    the analytic directional derivative (masks, 1/(1-p) factors, fused activation/residual/softmax dropouts) must
    match a central finite difference of the same stochastic function (a lost 1/(1-p) would be an 11 % error)."""
    z, meta = load_conf("conf_small")
    dev = _dev()
    m = _build(meta, conf_state(z), dropout=0.1).train()
    src = torch.from_numpy(z["src"]).to(dev)
    src_pos = torch.from_numpy(z["src_pos"]).to(dev)
    R = torch.from_numpy(z["R"]).to(dev)

    def loss():
        torch.manual_seed(99)
        return (m(src, src_pos) * R).sum()

    a = loss()
    b = loss()
    assert torch.equal(a.detach(), b.detach())
    m.zero_grad()
    a.backward()
    names = ["encoder.layers.0.ff_1.linear1.weight", "encoder.layers.0.attn.q_linear.weight",
             "encoder.layers.1.conv_module.pointwise_conv1.weight", "encoder.layers.1.ff_2.linear2.weight",
             "encoder.layers.0.attn.v_linear.weight"]
    params = dict(m.named_parameters())
    g = torch.Generator().manual_seed(5)
    for n in names:
        w = params[n]
        v = torch.randn(w.shape, generator=g).to(dev)
        analytic = float((w.grad * v).sum())
        eps = 2e-3
        with torch.no_grad():
            w.add_(eps * v)
            lp = float(loss())
            w.sub_(2 * eps * v)
            lm = float(loss())
            w.add_(eps * v)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - analytic) < 3e-2 * max(abs(analytic), 1.0), (n, fd, analytic)



@pytest.mark.parametrize("T,lens", [(7, [7, 7]), (11, [11, 5]), (45, [45, 0]), (64, [3, 64])])
def test_edge_lengths_vs_oracle(T, lens):
    """Edges of the subsampling and the key mask against the CPU restatement: 7 frames -> one encoder frame
    (T' = 1), 11 -> 2, an utterance with no valid frame at all (every key masked with -1e4: uniform attention, as the
    reference computes it) and one whose valid part ends before the first subsampled frame boundary."""
    from oracle import ref_conformer as RC
    dev = _dev()
    torch.manual_seed(T)
    ref = RC.TransformerEncoder(20, 16, 2, 2, 0.0)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if "norm" in k:
                p.add_(0.1 * torch.randn_like(p))
    B = len(lens)
    src = torch.randn(B, T, 20)
    src_pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        src[b, n:] = 0
        src_pos[b, :n] = torch.arange(1, n + 1)
    m = _build(dict(idim=20, d_model=16, N_e=2, heads=2), ref.state_dict())
    ref.train()
    m.train()
    yr = ref(src, src_pos)
    R = torch.randn_like(yr)
    (yr * R).sum().backward()
    y = m(src.to(dev), src_pos.to(dev))
    assert y.shape == yr.shape
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    (y * R.to(dev)).sum().backward()
    want = dict(ref.named_parameters())
    floor = 5e-2 * max(float(p.grad.abs().max()) for p in want.values())
    for k, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert grad_err(p.grad.cpu(), want[k].grad, floor) < 1e-3, k


@pytest.mark.parametrize("B,T,Cc", [(1, 1, 3), (2, 5, 70), (3, 40, 64), (2, 100, 130)])
def test_depthwise_conv_kernels_vs_torch(B, T, Cc):
    """asr_dwconv (forward, data gradient = flipped taps) and asr_dwconv_bwd_w + asr_colsum against
    torch.nn.functional.conv1d(groups = C, padding 15) in float64, including sequences shorter than the kernel."""
    import ctypes as C
    from asr_amd import _lib, ops
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(B * 100 + T)
    x, w, b, dy = torch.randn(B, T, Cc), torch.randn(Cc, 31), torch.randn(Cc), torch.randn(B, T, Cc)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.conv1d(xr.transpose(1, 2), wr.unsqueeze(1), br, padding=15, groups=Cc).transpose(1, 2)
    yr.backward(dy.double())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    xd, wd, bd, dyd = (t.to(dev).contiguous() for t in (x, w, b, dy))
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    _lib.check(lib.asr_dwconv(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, T, Cc, 31, 0, st))
    _lib.check(lib.asr_dwconv(dyd.data_ptr(), wd.data_ptr(), 0, dx.data_ptr(), B, T, Cc, 31, 1, st))
    part = torch.empty(int(lib.asr_dwconv_bwd_w_chunks(B, T)), 31 * Cc, device=dev)
    _lib.check(lib.asr_dwconv_bwd_w(dyd.data_ptr(), xd.data_ptr(), part.data_ptr(), B, T, Cc, 31, st))
    dwkc = torch.empty(31, Cc, device=dev)
    ops.colsum(part, dwkc.view(-1))
    assert rel_err(y.cpu(), yr.detach()) < 1e-5
    assert rel_err(dx.cpu(), xr.grad) < 1e-5
    assert rel_err(dwkc.t().cpu(), wr.grad) < 1e-5
    with pytest.raises(RuntimeError):
        _lib.check(lib.asr_dwconv(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, T, Cc, 15, 0, st))


def test_eval_mode_backward_vs_oracle():
    """Gradients in eval mode (BatchNorm normalises with its running statistics: the backward takes the branch
    without the batch-statistics terms) against the CPU restatement."""
    from oracle import ref_conformer as RC
    z, meta = load_conf("conf_small")
    dev = _dev()
    ref = RC.TransformerEncoder(meta["idim"], meta["d_model"], meta["N_e"], meta["heads"], 0.0)
    state = conf_state(z)
    for k in state:                                   # non-trivial running statistics
        if k.endswith("running_mean"):
            state[k] = 0.3 * torch.randn_like(state[k])
        if k.endswith("running_var"):
            state[k] = 0.5 + torch.rand_like(state[k])
    ref.load_state_dict(state)
    m = _build(meta, state)
    ref.eval()
    m.eval()
    src, src_pos = torch.from_numpy(z["src"]), torch.from_numpy(z["src_pos"])
    R = torch.from_numpy(z["R"])
    yr = ref(src, src_pos)
    (yr * R).sum().backward()
    y = m(src.to(dev), src_pos.to(dev))
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-4
    (y * R.to(dev)).sum().backward()
    want = dict(ref.named_parameters())
    floor = 5e-2 * max(float(p.grad.abs().max()) for p in want.values())
    for k, p in m.named_parameters():
        assert grad_err(p.grad.cpu(), want[k].grad, floor) < 1e-3, k
    for k, v in m.state_dict().items():               # eval mode leaves the buffers alone
        if "running_" in k:
            assert torch.equal(v.cpu(), state[k]), k


def test_full_size_config5_properties():
    """BASELINE config 5 at full size (the model bench.py's `config5_conformer` times: AttModel(encoder_type=
    'Conformer') = 16 layers, d_model 256, 4 heads, 48 M parameters; B=32, 1500 mel frames -> T'=374, V=1000,
    L=100, dropout 0.1 as the reference trains): far too slow for the CPU restatement, so -- like
    test_full_size_properties for configs 2 and 4 -- size-independent properties: finite loss and gradients for every
    parameter, and the derivative of the loss along a random parameter direction against a central finite difference.
    The dropout masks are drawn from torch's CPU generator (one seed per site), so re-seeding it before every pass
    makes the training-mode passes evaluate the same function.  The direction leaves out the two Conv2d weights of
    the subsampling: their gradient elements are sums over 227 392 patch rows with random signs, whose fp32
    round-off alone moves the directional derivative by ~10 % (tools/cnn_check.py: against a float64 torch
    reference the kernels' conv.2.weight gradient is within 1.7e-3 of the largest element -- closer than torch's own
    fp32 convolution backward, 4.5e-3 -- and tools/fd_config5.py shows every other parameter group converging to
    the analytic value); they are covered by the golden cases and by tools/cnn_check.py instead."""
    import types
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    dev = _dev()
    B, Tm, V, L = 32, 1500, 1000, 100
    hp = types.SimpleNamespace(lmfb_dim=80, frame_stacking=1, num_hidden_nodes=320, num_hidden_nodes_encoder=256,
                               num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.2,
                               encoder_type="Conformer", decoder_type="Attention", num_classes=V, eos_id=1,
                               T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=200,
                               score_func="log_softmax")
    torch.manual_seed(1234)
    m = AttModel(hp).to(dev).train()
    assert sum(p.numel() for p in m.parameters()) > 40e6
    g = torch.Generator().manual_seed(1234)
    lens = torch.linspace(1500, 900, B).long()
    x = torch.randn(B, Tm, 80, generator=g)
    src_pos = torch.zeros(B, Tm, dtype=torch.int64)
    for b in range(B):
        x[b, lens[b]:] = 0
        src_pos[b, :lens[b]] = torch.arange(1, int(lens[b]) + 1)
    tl = torch.linspace(100, 50, B).long()
    tgt = torch.randint(0, V, (B, L), generator=g)
    for b in range(B):
        tgt[b, tl[b]:] = 0
    x, src_pos, tgt, tl_d = x.to(dev), src_pos.to(dev), tgt.to(dev), tl.to(dev)

    def loss_of():
        torch.manual_seed(99)          # same dropout masks in every pass
        return label_smoothing_loss(m(x, lens, tgt, src_pos), tgt, tl_d, hp.T_norm, hp.B_norm)

    loss = loss_of()
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    with torch.no_grad():
        torch.manual_seed(7)
        dirs = {k: (torch.zeros_like(p) if k in ("encoder.cnn_encoder.conv.0.weight", "encoder.cnn_encoder.conv.2.weight")
                    else torch.randn_like(p)) for k, p in m.named_parameters()}
        analytic = sum(float((p.grad.double() * dirs[k].double()).sum()) for k, p in m.named_parameters())
        # 48 M parameters all move at once: the step must be small for the loss to be quadratic along it (the
        # central difference at 1e-3 is 14 % off, at 2.5e-4 1 %); two steps, Richardson-extrapolated
        fds = []
        for eps in (5e-4, 2.5e-4):
            vals = []
            for sgn in (+1, -1):
                for k, p in m.named_parameters():
                    p.add_(dirs[k], alpha=sgn * eps)
                vals.append(float(loss_of()))
                for k, p in m.named_parameters():
                    p.add_(dirs[k], alpha=-sgn * eps)
            fds.append((vals[0] - vals[1]) / (2 * eps))
        fd = fds[1] + (fds[1] - fds[0]) / 3.0
    assert abs(fd - analytic) / max(abs(analytic), 1e-6) < 3e-2, (fds, fd, analytic)


def test_weight_gradients_on_side_stream_match_one_stream():
    """The blocks' weight / bias gradients run on the side stream when a stream-aware owner has enabled it
    (asr_amd/streams.py; the bench's config-5 leg, BucketedGradReducer): same seeds, every gradient against the
    one-stream pass (split-K products add with float atomics: equal to rounding, not bit for bit), three passes in a
    row so that a block recycled under a pending side-stream read would show."""
    from asr_amd import streams
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    dev = _dev()
    hp = R.make_hp(lmfb_dim=40, frame_stacking=1, num_hidden_nodes_encoder=256, num_hidden_nodes_decoder=32,
                   num_classes=50, encoder_type="Conformer", encoder_dropout=0.1)
    torch.manual_seed(11)
    m = AttModel(hp)
    m.encoder.encoder.layers = torch.nn.Sequential(*list(m.encoder.encoder.layers)[:3])
    m = m.to(dev).train()
    B, T, L = 4, 300, 12
    lens = torch.tensor([300, 251, 212, 180])
    x = torch.randn(B, T, 40)
    src_pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens.tolist()):
        x[b, n:] = 0
        src_pos[b, :n] = torch.arange(1, n + 1)
    tl = torch.tensor([12, 9, 7, 5])
    tgt = torch.randint(1, 50, (B, L))
    x, src_pos, tgt, tl = x.to(dev), src_pos.to(dev), tgt.to(dev), tl.to(dev)

    def grads(side):
        streams.enable(side)
        torch.manual_seed(7)                      # the dropout seeds come from torch's generator
        m.zero_grad(set_to_none=True)
        loss = label_smoothing_loss(m(x, lens, tgt, src_pos), tgt, tl)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.clone() for n, p in m.named_parameters()}

    was = streams.ENABLED
    try:
        l0, g0 = grads(False)
        for _ in range(3):
            l1, g1 = grads(True)
            assert l1 == l0
            for n in g0:
                assert torch.isfinite(g1[n]).all(), n
                assert float((g1[n] - g0[n]).abs().max()) <= 1e-5 * float(g0[n].abs().max()) + 1e-12, n
    finally:
        streams.enable(was)


@pytest.mark.parametrize("n", [1, 10, 4096, 4100, 3 * 11968 * 64])
@pytest.mark.parametrize("masked", [False, True])
def test_scale_mask_scalar_and_16_byte_paths(n, masked):
    """asr_scale_mask (a block's output dropout backward): the 16-byte kernel (n % 4 == 0, aligned) and the scalar one,
    with and without a mask, bit for bit against torch; a misaligned view falls back to the scalar kernel."""
    import ctypes as C
    from asr_amd import _lib
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shift in (0, 1):                                        # shift 1: pointers 4 bytes off a 16-byte boundary
        dy = torch.randn(n + shift, generator=g).to(dev)[shift:]
        mask = (torch.rand(n + shift, generator=g) < 0.7).to(torch.uint8).to(dev)[shift:] if masked else None
        out = torch.full((n + shift,), 7.0, device=dev)[shift:]
        assert lib.asr_scale_mask(dy.data_ptr(), mask.data_ptr() if masked else 0, out.data_ptr(), n, C.c_float(1.25), st) == 0
        want = dy * 1.25 * (mask.float() if masked else 1.0)
        assert torch.equal(out, want)

