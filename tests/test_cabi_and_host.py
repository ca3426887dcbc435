"""CPU: the C-ABI library loads and exports every symbol include/asr_amd.h declares
(no compute calls without a GPU), and the host-side mirror keeps the reference's
interface (names, state_dict, init, error behaviour)."""
import os
import re

import pytest
import torch

from tests._util import CASES, load_case, params_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "asr_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(asr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from asr_amd import _lib
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names, set(_lib.SYMBOLS) ^ set(names)
    assert lib.asr_version() >= 100


def test_bad_arguments_are_reported_not_thrown():
    from asr_amd import _lib
    lib = _lib.load()
    rc = lib.asr_ls_ce_fwd_bwd(0, 0, 0, 1, 1, 2, 0.1, 1, 0, 0, 0, 0, 0)
    assert rc == -1
    assert b"null" in lib.asr_last_error()
    rc = lib.asr_blstm_fwd(0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0)
    assert rc == -1
    with pytest.raises(RuntimeError):
        _lib.check(rc)


@pytest.mark.parametrize("name", CASES)
def test_state_dict_is_the_references(name):
    from asr_amd.Models.AttModel import AttModel
    z, hp = load_case(name)
    m = AttModel(hp)
    want = {k[len("param."):]: z[k].shape for k in z.files if k.startswith("param.")}
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    m.load_state_dict(params_of(z))       # strict


def test_seeded_init_equals_oracle_init():
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=16, num_encoder_layer=2, lmfb_dim=4, num_classes=11)
    torch.manual_seed(3)
    a = AttModel(hp).state_dict()
    torch.manual_seed(3)
    b = R.AttModel(hp).state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_no_cpu_fallback():
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from oracle import ref_model as R
    hp = R.make_hp(num_hidden_nodes=8, num_encoder_layer=1, lmfb_dim=4, num_classes=5, frame_stacking=1)
    m = AttModel(hp)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 4), torch.tensor([3]), torch.zeros(1, 2, dtype=torch.long))
    with pytest.raises(RuntimeError):
        label_smoothing_loss(torch.zeros(1, 2, 5), torch.zeros(1, 2, dtype=torch.long), torch.tensor([2]))
    with pytest.raises(NotImplementedError):
        AttModel(R.make_hp(encoder_type="Wave"))
    with pytest.raises(NotImplementedError):      # Models/encoder.py:20-22: the VGG front-end, chosen only without frame stacking
        AttModel(R.make_hp(encoder_type="CNN", frame_stacking=1, num_hidden_nodes=8, num_encoder_layer=1, lmfb_dim=4))
    AttModel(R.make_hp(encoder_type="CNN", frame_stacking=3, num_hidden_nodes=8, num_encoder_layer=1, lmfb_dim=4))   # :18-19
    # the Conformer branch (Models/AttModel.py:16-17): same module tree as the reference, no CPU path
    from asr_amd.Models.Transformer_encoder import TransformerEncoder
    enc = TransformerEncoder(20, 16, 1, 2, 0.0)
    assert len(enc.state_dict()) == 45
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 30, 20), torch.ones(1, 30, dtype=torch.long))


def test_frame_stacking_matches_oracle():
    from asr_amd.utils.utils import frame_stacking
    from oracle import ref_model as R
    x = torch.randn(3, 17, 4)
    lens = torch.tensor([17, 9, 4])
    a, al = frame_stacking(x, lens, 3)
    b, bl = R.frame_stacking(x, lens, 3)
    assert torch.equal(a, b) and torch.equal(al, bl) and a.shape == (3, 5, 12)


def test_target_must_be_contiguous_like_the_reference():
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    t = torch.zeros(2, 6, dtype=torch.long)[:, :4]
    with pytest.raises(RuntimeError, match="contiguous"):
        label_smoothing_loss(torch.zeros(2, 4, 5), t, torch.tensor([4, 4]))


def test_host_side_dispatch_queries_and_split_k_rule():
    """Host logic that needs no GPU: which backward form reads the transposed W_hh, the split-K rule of the weight
    gradients (one resident round of 512 workgroups on 128x128 tiles), the readable span of GEMM operands."""
    import torch
    from asr_amd import _lib, ops
    lib = _lib.load()
    # config-2 layer, batch-major output strides: reduce-scatter form, W_hh only
    assert lib.asr_blstm_bwd_needs_transposed(500, 32, 320, 640, 500 * 640, 1) == 0
    assert lib.asr_blstm_bwd_needs_transposed(500, 32, 320, 640, 500 * 640, 0) == 1     # no workspace: step-wise kernels
    assert lib.asr_blstm_bwd_needs_transposed(500, 512, 320, 640, 500 * 640, 1) == 1    # too many batch tiles for one launch
    assert ops.pick_splitk(2048, 256, 11968) == 16          # Conformer ff dW: 32 tiles x 16 = 512 workgroups
    assert ops.pick_splitk(2560, 640, 16000) == 5           # config-2 dW_ih, both directions: 100 tiles x 5
    assert ops.pick_splitk(16000, 2560, 640) == 1
    assert 1 <= ops.pick_splitk(256, 256, 11968) <= 64
    t = torch.zeros(10, 8)
    assert ops._span(t, 0, 10, 8, 0, 8, 0, 1, 0, 0, 0, False) == 80 * 4
    assert ops._span(t, 16, 8, 8, 0, 8, 0, 1, 0, 0, 0, False) == 64 * 4
    assert ops._span(t, 0, 11, 8, 0, 8, 0, 1, 0, 0, 0, False) == 0            # a row past the tensor
    assert ops._span(t, 0, 8, 8, 4, 8, 1000, 1, 0, 0, 0, False) == 0          # two-level view into another allocation
    assert ops._span(t, 0, 8, 8, 4, 8, 32, 1, 0, 0, 0, False) == 80 * 4       # two-level view inside the tensor
    assert ops._span(t.t(), 0, 8, 10, 0, 10, 0, 1, 0, 0, 0, False) == 0       # not contiguous
