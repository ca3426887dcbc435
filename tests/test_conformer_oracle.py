"""CPU: the Conformer-encoder restatement (oracle/ref_conformer.py) vs the golden vectors produced by
the reference's Models/Transformer_encoder.py (oracle/make_golden_conformer.py)."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ref_conformer as RC
from tests._util import GOLDEN_DIR, rel_err

CONF_CASES = sorted(os.path.splitext(os.path.basename(p))[0]
                    for p in glob.glob(os.path.join(GOLDEN_DIR, "conformer", "*.npz")))


def load_conf(name):
    z = np.load(os.path.join(GOLDEN_DIR, "conformer", name + ".npz"), allow_pickle=False)
    return z, dict(ast.literal_eval(str(z["meta"])))


def conf_state(z, dtype=torch.float32):
    out = {}
    for k in z.files:
        if k.startswith("param."):
            t = torch.from_numpy(z[k])
            out[k[len("param."):]] = t.to(dtype) if t.is_floating_point() else t
    return out


def grad_floor(z, tag):
    """Some gradients are zero in exact arithmetic (a bias in front of BatchNorm, the key bias in
    front of a softmax): rounding noise of 1e-7 there is judged against 5 % of the largest
    gradient of the model, not against itself."""
    return 5e-2 * max(float(np.abs(z[k]).max()) for k in z.files if k.startswith(f"{tag}.grad."))


def grad_err(got, want, floor):
    got = torch.as_tensor(got, dtype=torch.float64)
    want = torch.as_tensor(want, dtype=torch.float64)
    return float((got - want).abs().max()) / max(float(want.abs().max()), floor)


def build(meta, z, dtype):
    m = RC.TransformerEncoder(meta["idim"], meta["d_model"], meta["N_e"], meta["heads"], 0.0).to(dtype)
    m.load_state_dict(conf_state(z, dtype))
    return m


def test_have_cases():
    assert len(CONF_CASES) >= 3


@pytest.mark.parametrize("name", CONF_CASES)
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_conformer_matches_reference(name, tag, dtype, tol):
    z, meta = load_conf(name)
    m = build(meta, z, dtype)
    assert sorted(m.state_dict().keys()) == sorted(k[len("param."):] for k in z.files if k.startswith("param."))
    src = torch.from_numpy(z["src"]).to(dtype)
    src_pos = torch.from_numpy(z["src_pos"])
    m.eval()
    with torch.no_grad():
        assert rel_err(m(src, src_pos), z[f"{tag}.eval_out"]) < tol
    m.train()
    y = m(src, src_pos)
    assert rel_err(y.detach(), z[f"{tag}.train_out"]) < tol
    (y * torch.from_numpy(z["R"]).to(dtype)).sum().backward()
    n = 0
    floor = grad_floor(z, tag)
    for k, p in m.named_parameters():
        if f"{tag}.grad.{k}" in z.files:
            assert grad_err(p.grad, z[f"{tag}.grad.{k}"], floor) < 50 * tol, k
            n += 1
    assert n > 20
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v, z[f"{tag}.after.{k}"]) < tol, k
        if "num_batches" in k:
            assert int(v) == int(z[f"{tag}.after.{k}"])


def test_rel_shift_index_map():
    """rel_shift (Transformer_encoder.py:110-121) equals the gather the kernels use."""
    T = 7
    x = torch.randn(2, 3, T, T)
    r, c = RC.rel_shift_index(T)
    want = RC.rel_shift(x)
    got = torch.where(c == 0, torch.zeros(()), x[:, :, r, (c - 1).clamp(min=0)])
    assert torch.equal(got, want)
