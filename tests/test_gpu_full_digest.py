"""GPU: the HIP path at the HEADLINE size (BASELINE config 2: B=32, T=500, L=100, V=1000, 5 x BLSTM 320) against
digests of the REFERENCE's own step at that size (tests/golden/full/*.npz, written by oracle/make_golden_full.py by
importing /root/reference in the build container; what a digest holds: tests/_full_digest.py).

Two cases -- the default initialisation (what bench.py times) and a rescaled model whose alignment is peaked, so
that the attention's gradients are not degenerate -- each in the library's default arithmetic (bf16x3 products)
and with exact fp32 products, against the reference in float64 (and, for the loss and outputs, in float32 as the
reference computes it).  Bars (north star): loss 1e-4 relative, youtput 1e-4 of its largest entry, gradients 1e-3:
per tensor, the L2 norm, the largest entry and 8 random projections of the difference (each ~ the L2 error of the
whole tensor) against 1e-3 of the tensor's own L2 norm (resp. largest entry), with a floor of 1e-3 of the model's
largest such figure for tensors whose gradient is zero up to round-off at random initialisation.
"""
import os

import numpy as np
import pytest
import torch

from tests import _full_digest as D

pytestmark = pytest.mark.gpu

FULL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full")
LOSS_TOL, OUT_TOL, GRAD_TOL = 1e-4, 1e-4, 1e-3


def _hip_digest(case, exact):
    from asr_amd import _lib
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    from asr_amd.utils.utils import frame_stacking
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    dev = torch.device("cuda:0")
    lib = _lib.load()
    hp = D.make_hp()
    torch.manual_seed(D.SEED)
    m = AttModel(hp)                                     # same seeded default initialisation as the reference's
    m.load_state_dict(D.scaled_state(m.state_dict(), case))
    m = m.to(dev).train()
    mel, mel_len, tgt, text_len = D.make_inputs(hp)
    x, lens = frame_stacking(mel, mel_len, hp.frame_stacking)
    x = x.to(dev).requires_grad_(True)
    og, ob = lib.asr_gemm_precision(-1), lib.asr_blstm_precision(-1)
    try:
        if exact:
            lib.asr_gemm_precision(0)
            lib.asr_blstm_precision(0)
        y = m(x, lens, tgt.to(dev))
        loss = label_smoothing_loss(y, tgt.to(dev), text_len.to(dev), hp.T_norm, hp.B_norm)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        lib.asr_gemm_precision(og)
        lib.asr_blstm_precision(ob)
    names = [k for k, _ in m.named_parameters()]
    ps = [p for _, p in m.named_parameters()]
    return D.step_digest(names, ps, [p.grad for p in ps], y, loss.detach(), x.grad)


@pytest.mark.parametrize("exact", [False, True], ids=["default-bf16x3", "exact-f32"])
@pytest.mark.parametrize("case", ["init", "peaked"])
def test_config2_full_size_vs_reference_digest(case, exact):
    z = np.load(os.path.join(FULL_DIR, f"config2_{case}.npz"), allow_pickle=False)
    got = _hip_digest(case, exact)
    names = str(z["f32.names"]).split("\n")
    assert str(got["names"]).split("\n") == names
    # same weights as the reference had (seeded default initialisation + the case's scaling)
    for i, n in enumerate(names):
        want = z[f"f32.param.{i}"]
        assert np.allclose(got[f"param.{i}"], want, rtol=1e-6, atol=1e-6 * max(1.0, abs(want[1]))), n
    # loss and outputs: against the reference as it computes (float32) and promoted to float64
    for tag in ("f32", "f64"):
        want_loss = float(z[f"{tag}.loss"])
        assert abs(float(got["loss"]) - want_loss) < LOSS_TOL * abs(want_loss), (tag, float(got["loss"]), want_loss)
        ymax = float(z[f"{tag}.y.stat"][1])
        assert float(np.abs(got["y.sample"] - z[f"{tag}.y.sample"]).max()) < OUT_TOL * ymax, tag
        assert abs(float(got["y.stat"][0]) - float(z[f"{tag}.y.stat"][0])) < OUT_TOL * float(z[f"{tag}.y.stat"][0])
    # gradients against the float64 reference
    keys = [f"grad.{i}" for i in range(len(names))] + ["dx"]
    ref = {k: z[f"f64.{k}"] for k in keys}
    floor_l2 = 1e-3 * max(float(ref[k][0]) for k in keys[:-1])
    floor_mx = 1e-3 * max(float(ref[k][1]) for k in keys[:-1])
    worst = (0.0, None)
    for k, n in zip(keys, names + ["d loss / d x"]):
        g, r = got[k], ref[k]
        l2, mx = max(float(r[0]), floor_l2), max(float(r[1]), floor_mx)
        e_norm = abs(float(g[0]) - float(r[0])) / l2
        e_max = abs(float(g[1]) - float(r[1])) / mx
        e_proj = float(np.abs(g[2:] - r[2:]).max()) / l2
        worst = max(worst, (max(e_norm, e_max, e_proj), n))
        assert e_norm < GRAD_TOL and e_max < GRAD_TOL and e_proj < GRAD_TOL, (n, e_norm, e_max, e_proj)
    if case == "peaked":
        # the point of the case: the attention's gradients are not degenerate (every tensor's largest entry is
        # within 1e-3 of the model's largest), so the bars above held WITHOUT the floor
        att = [i for i, n in enumerate(names) if ".att." in n]
        assert all(float(ref[f"grad.{i}"][1]) > floor_mx for i in att), [(names[i], float(ref[f"grad.{i}"][1])) for i in att]
    print(f"config2_{case} {'exact f32' if exact else 'bf16x3'}: worst gradient figure {worst[0]:.2e} ({worst[1]})")
