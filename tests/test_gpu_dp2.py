"""GPU, SURVEY.md section 4 "distributed" tier on a one-GPU box: the same global batch once on one rank
and once split over TWO ranks (two processes sharing GPU 0, gloo between them, the reducer's
host-staged path).  The loss is a sum over utterances, so the two ranks' gradients add up to the
single-rank ones; the reducer averages like DDP (train.py:141), so 2 x mean must equal them."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_case():
    from oracle import ref_model as R
    hp = R.make_hp(lmfb_dim=12, frame_stacking=1, num_hidden_nodes=32, num_encoder_layer=3, num_classes=21,
                   encoder_dropout=0.0)
    g = torch.Generator().manual_seed(77)
    from asr_amd.Models.AttModel import AttModel
    torch.manual_seed(77)
    state = {k: v.clone() for k, v in AttModel(hp).state_dict().items()}
    B, T, L = 4, 48, 5
    lengths = torch.tensor([48, 40, 48, 17])          # each half keeps one full-length utterance
    x = torch.randn(B, T, 12, generator=g)
    for b in range(B):
        x[b, lengths[b]:] = 0
    targets = torch.randint(0, 21, (B, L), generator=g)
    tl = torch.tensor([5, 3, 4, 5])
    return hp, state, x, lengths, targets, tl


def test_two_rank_gradients_equal_single_rank(tmp_path):
    from asr_amd.Loss.label_smoothing import label_smoothing_loss
    from asr_amd.Models.AttModel import AttModel
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    dev = torch.device("cuda:0")
    hp, state, x, lengths, targets, tl = make_case()
    m = AttModel(hp)
    m.load_state_dict(state)
    m = m.to(dev).train()
    loss = label_smoothing_loss(m(x.to(dev), lengths, targets.to(dev)), targets.to(dev), tl.to(dev), hp.T_norm,
                                hp.B_norm)
    loss.backward()
    torch.cuda.synchronize()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp2_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=110)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank of the two-rank run did not finish in 110 s")
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert abs((float(r0["loss"]) + float(r1["loss"])) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    # 1e-6 of the tensor's largest entry; a gradient that is (nearly) zero in exact arithmetic -- the bias in front
    # of the energy's tanh sums to ~1e-7 by cancellation -- is judged against 1e-4 of the model's largest gradient
    floor = 1e-4 * max(float(p.grad.abs().max()) for p in m.parameters())
    for k, p in m.named_parameters():
        assert np.array_equal(r0[k], r1[k]), k                       # every rank holds the same mean
        err = float((2.0 * torch.from_numpy(r0[k]) - p.grad.cpu()).abs().max())
        assert err < 1e-6 * max(float(p.grad.abs().max()), floor), (k, err)
