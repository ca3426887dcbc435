"""CPU: the CTC oracle (oracle/ref_ctc.py) vs the golden vectors produced by the reference's
CTCModel + train.py:56-57, and torch's F.ctc_loss vs the textbook recursion."""
import pytest
import torch

from oracle import ref_ctc as RC
from tests._util import CTC_CASES, load_case, params_of, rel_err


def test_fixtures_present():
    assert len(CTC_CASES) >= 4


@pytest.mark.parametrize("name", CTC_CASES)
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_ctc_forward_backward_matches_reference(name, tag, dtype, tol):
    z, hp = load_case(name)
    m = RC.CTCModel(hp).to(dtype)
    m.load_state_dict(params_of(z, dtype))
    m.train()
    x = torch.from_numpy(z["x"]).to(dtype).requires_grad_(True)
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"])
    tl = torch.from_numpy(z["text_lengths"])
    y = m(x, lengths, targets)
    y.retain_grad()
    loss = RC.ctc_loss_train(y, targets, lengths, tl)
    loss.backward()
    assert rel_err(y.detach(), z[f"{tag}.youtput"]) < tol
    assert abs(float(loss) - float(z[f"{tag}.loss"])) / abs(float(z[f"{tag}.loss"])) < tol
    assert rel_err(y.grad, z[f"{tag}.dyoutput"]) < 50 * tol
    assert rel_err(x.grad, z[f"{tag}.dx"]) < 50 * tol
    for k, p in m.named_parameters():
        assert rel_err(p.grad, z[f"{tag}.grad.{k}"]) < 50 * tol, k


@pytest.mark.parametrize("name", CTC_CASES)
def test_ctc_state_dict_and_decode(name):
    z, hp = load_case(name)
    m = RC.CTCModel(hp)
    want = sorted(k[len("param."):] for k in z.files if k.startswith("param."))
    assert sorted(m.state_dict().keys()) == want
    assert "decoder.weight" in want and "decoder.bias" in want
    m.load_state_dict(params_of(z))
    m.eval()
    got = m.decode(torch.from_numpy(z["x"]), torch.from_numpy(z["lengths"]))
    ref = [[int(v) for v in row if v >= 0] for row in z["decode"]]
    assert got == ref


@pytest.mark.parametrize("name", CTC_CASES)
def test_torch_ctc_equals_textbook_recursion(name):
    """Pins the third-party arithmetic: F.ctc_loss (what train.py:57 calls) against the
    published alpha recursion, on the reference's own logits."""
    z, hp = load_case(name)
    y = torch.from_numpy(z["f64.youtput"])
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"])
    tl = torch.from_numpy(z["text_lengths"])
    nll = RC.ctc_nll_explicit(y, targets, lengths, tl)
    lp = torch.log_softmax(y, 2).transpose(0, 1)
    want = torch.nn.functional.ctc_loss(lp, targets, lengths, tl, blank=0, reduction="none")
    assert rel_err(nll, want) < 1e-10
    mean = (nll / tl.clamp_min(1)).mean()
    assert abs(float(mean) - float(z["f64.loss"])) < 1e-10 * abs(float(z["f64.loss"]))


# ------------------------------------------------ host mirror (no GPU needed) --
@pytest.mark.parametrize("name", CTC_CASES)
def test_ctc_mirror_state_dict_is_the_references(name):
    from asr_amd.Models.CTCModel import CTCModel
    z, hp = load_case(name)
    m = CTCModel(hp)
    want = {k[len("param."):]: tuple(z[k].shape) for k in z.files if k.startswith("param.")}
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    m.load_state_dict(params_of(z))       # strict


def test_ctc_mirror_seeded_init_equals_oracle_init():
    from asr_amd.Models.CTCModel import CTCModel
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=16, num_encoder_layer=2, lmfb_dim=4, num_classes=11, decoder_type="CTC")
    torch.manual_seed(3)
    a = CTCModel(hp).state_dict()
    torch.manual_seed(3)
    b = RC.CTCModel(hp).state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_ctc_mirror_has_no_cpu_fallback():
    from asr_amd.Loss.ctc import ctc_loss
    from asr_amd.Models.CTCModel import CTCModel
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=8, num_encoder_layer=1, lmfb_dim=4, num_classes=5, frame_stacking=1)
    m = CTCModel(hp)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 4), torch.tensor([3]))
    with pytest.raises(RuntimeError):
        ctc_loss(torch.zeros(1, 3, 5), torch.ones(1, 2, dtype=torch.long), torch.tensor([3]), torch.tensor([2]))
    hp.encoder_type = "Wave"
    with pytest.raises(NotImplementedError):
        CTCModel(hp)
