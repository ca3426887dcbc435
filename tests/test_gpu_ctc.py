"""GPU parity tests of the CTC head: asr_amd.Models.CTCModel + asr_amd.Loss.ctc
through the C ABI against the reference's golden vectors and the CPU oracle.
Tolerances as for the attention path: loss <= 1e-4, logits <= 1e-4, gradients <= 1e-3
(scale-relative max error); decoded label sequences identical."""
import pytest
import torch

from tests._util import CTC_CASES, load_case, params_of, rel_err

pytestmark = pytest.mark.gpu

LOSS_TOL, OUT_TOL, GRAD_TOL = 1e-4, 1e-4, 1e-3


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU: refusing to pass on a fallback"
    return torch.device("cuda:0")


def _cpu_ctc(z, tgt, il, tl):
    """train.py:56-57 on the CPU in float64 (the oracle), loss + per-utterance nll + d/dz."""
    from oracle import ref_ctc as RC
    zz = z.double().requires_grad_(True)
    loss = RC.ctc_loss_train(zz, tgt, il, tl)
    g, = torch.autograd.grad(loss, zz)
    lp = torch.log_softmax(z.double(), 2).transpose(0, 1)
    nll = torch.nn.functional.ctc_loss(lp, tgt, il, tl, blank=0, reduction="none")
    return float(loss.detach()), nll, g


@pytest.mark.parametrize("B,T,V,L,hi,seed", [
    (3, 20, 11, 5, 11, 1),       # plain
    (4, 30, 5, 7, 3, 2),         # labels in {1,2}: repeats everywhere, skip transitions forbidden
    (2, 9, 7, 4, 7, 3),          # barely feasible lengths
    (5, 64, 1000, 20, 1000, 4),  # vector path (V % 4 == 0)
    (3, 400, 37, 150, 37, 5),    # long label rows (S = 301 > 256 threads of the gradient kernel)
    (2, 12, 6, 0, 6, 6),         # L = 0: blank-only paths
    (32, 500, 1000, 100, 1000, 7),   # config-2 shape
])
def test_ctc_kernel_vs_oracle(B, T, V, L, hi, seed):
    from asr_amd.Loss.ctc import ctc_loss
    dev = _dev()
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, T, V, generator=g) * 2
    tgt = torch.randint(1, max(hi, 2), (B, L), generator=g) if L > 0 else torch.zeros(B, 0, dtype=torch.int64)
    tl = torch.randint(0, L + 1, (B,), generator=g)
    tl[0] = L
    # every utterance long enough for its labels (repeats need a blank between them)
    need = torch.tensor([int(tl[b]) + int((tgt[b, 1:tl[b]] == tgt[b, :max(int(tl[b]) - 1, 0)]).sum()) for b in range(B)])
    il = torch.maximum(torch.randint(1, T + 1, (B,), generator=g), need).clamp(max=T)
    il[0] = T
    assert bool((il >= need).all())
    for b in range(B):
        tgt[b, tl[b]:] = 0
    want_loss, want_nll, want_g = _cpu_ctc(z, tgt, il, tl)
    zd = z.to(dev).requires_grad_(True)
    loss, nll = ctc_loss(zd, tgt.to(dev), il, tl.to(dev), blank=0, return_nll=True)   # mel_lengths on the CPU, as train.py has them
    loss.backward()
    assert abs(float(loss) - want_loss) < LOSS_TOL * abs(want_loss)
    assert rel_err(nll.cpu(), want_nll) < LOSS_TOL
    assert rel_err(zd.grad.cpu(), want_g) < GRAD_TOL
    for b in range(B):    # frames past the input length: exactly zero gradient
        assert float(zd.grad[b, int(il[b]):].abs().max() if il[b] < T else 0.0) == 0.0
    # bitwise reproducible (no float atomics)
    zd2 = z.to(dev).requires_grad_(True)
    loss2 = ctc_loss(zd2, tgt.to(dev), il.to(dev), tl.to(dev))
    loss2.backward()
    assert torch.equal(loss2, loss) and torch.equal(zd2.grad, zd.grad)


def test_ctc_infeasible_and_errors():
    from asr_amd.Loss.ctc import ctc_loss
    dev = _dev()
    z = torch.randn(2, 3, 5, device=dev)
    tgt = torch.tensor([[1, 1, 2, 2], [1, 2, 3, 4]], device=dev)
    loss, nll = ctc_loss(z, tgt, torch.tensor([3, 3]), torch.tensor([4, 2]), return_nll=True)
    assert torch.isinf(nll[0]) and torch.isfinite(nll[1]) and torch.isinf(loss)     # as F.ctc_loss(zero_infinity=False)
    with pytest.raises(RuntimeError):
        ctc_loss(z.cpu(), tgt, torch.tensor([3, 3]), torch.tensor([4, 2]))
    with pytest.raises(RuntimeError):
        ctc_loss(torch.randn(1, 4, 5, device=dev), torch.ones(1, 600, dtype=torch.int64, device=dev),
                 torch.tensor([4]), torch.tensor([1]))


@pytest.mark.parametrize("name", CTC_CASES)
def test_ctc_model_golden(name):
    from asr_amd.Loss.ctc import ctc_loss
    from asr_amd.Models.CTCModel import CTCModel
    dev = _dev()
    z, hp = load_case(name)
    m = CTCModel(hp)
    m.load_state_dict(params_of(z))
    m = m.to(dev).train()
    x = torch.from_numpy(z["x"]).to(dev).requires_grad_(True)
    lengths = torch.from_numpy(z["lengths"])                 # CPU, as the reference passes them
    targets = torch.from_numpy(z["targets"]).to(dev)
    tl = torch.from_numpy(z["text_lengths"]).to(dev)
    y = m(x, lengths, targets, None)
    y.retain_grad()
    loss = ctc_loss(y, targets, lengths, tl, blank=0)
    loss.backward()
    assert rel_err(y.detach().cpu(), z["f64.youtput"]) < OUT_TOL
    assert abs(float(loss) - float(z["f64.loss"])) < LOSS_TOL * abs(float(z["f64.loss"]))
    assert rel_err(y.grad.cpu(), z["f64.dyoutput"]) < GRAD_TOL
    assert rel_err(x.grad.cpu(), z["f64.dx"]) < GRAD_TOL
    names = [k for k, _ in m.named_parameters()]
    assert sorted(names) == sorted(k[len("f64.grad."):] for k in z.files if k.startswith("f64.grad."))
    for k, p in m.named_parameters():
        assert rel_err(p.grad.cpu(), z[f"f64.grad.{k}"]) < GRAD_TOL, k
    # greedy decoding: identical label sequences
    m.eval()
    got = m.decode(x.detach(), lengths)
    assert got == [[int(v) for v in row if v >= 0] for row in z["decode"]]


def test_argmax_rows_vs_torch():
    import ctypes as C
    from asr_amd import _lib
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    for rows, cols in ((1, 1), (7, 37), (130, 1000), (5, 7001)):
        x = torch.randn(rows, cols, generator=g)
        x[0, : min(cols, 3)] = 9.0                  # a tie: the first maximum wins
        xd = x.to(dev)
        out = torch.empty(rows, dtype=torch.int64, device=dev)
        assert lib.asr_argmax_rows(xd.data_ptr(), rows, cols, cols, out.data_ptr(),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        assert torch.equal(out.cpu(), x.argmax(dim=1))
