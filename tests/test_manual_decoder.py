"""CPU: the closed-form decoder BPTT (what the HIP kernels implement) vs autograd."""
import pytest
import torch

from oracle import manual_decoder as MD
from oracle import ref_model as R
from tests._util import CASES, load_case, params_of, rel_err


@pytest.mark.parametrize("name", CASES)
def test_manual_decoder_matches_autograd(name):
    z, hp = load_case(name)
    m = R.AttModel(hp).double()
    m.load_state_dict(params_of(z, torch.float64))
    x = torch.from_numpy(z["x"]).double()
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"])
    with torch.no_grad():
        hb = m.encoder(x, lengths)
    hb = hb.clone().requires_grad_(True)
    y = m.decoder(hb, lengths, targets)
    torch.manual_seed(0)
    dY = torch.randn_like(y)
    (y * dY).sum().backward()

    P = {k: v.detach() for k, v in m.decoder.named_parameters()}
    Y, sv = MD.forward(P, hb.detach(), lengths, targets)
    assert rel_err(Y, y.detach()) < 1e-12
    dh, G = MD.backward(P, hb.detach(), targets, sv, dY)
    assert rel_err(dh, hb.grad) < 1e-10
    for k, p in m.decoder.named_parameters():
        assert rel_err(G[k], p.grad) < 1e-10, k
