"""Digest of one full-size training step (BASELINE config 2: B=32, T=500, L=100, V=1000, 5 x BLSTM 320).

A step at this size produces 12.8 MB of youtput and 60 MB of gradients: too much to commit as a fixture.  The
digest keeps, per case, the loss, 4096 sampled youtput entries, and for every gradient tensor (and d loss / d x) its
L2 norm, its largest entry and NPROJ projections on fixed pseudo-random directions -- a few KB.  A projection of the
DIFFERENCE of two tensors on a N(0, 1) direction has standard deviation |difference|_2, so comparing projections
bounds the L2 error of the whole tensor without shipping it.

Used by oracle/make_golden_full.py (which runs THE REFERENCE in the build container and writes
tests/golden/full/*.npz) and by tests/test_gpu_full_digest.py (which recomputes the same digest from the HIP
path).  Everything here is deterministic given the seeds: the weights are torch's default initialisation under
torch.manual_seed(SEED) -- the reference's AttModel and asr_amd's draw identical values, and the digest carries a
checksum of every parameter so that this is verified, not assumed -- optionally rescaled by `peaked_scale`.
"""
import types

import numpy as np
import torch

SEED = 1234
NPROJ = 8
NSAMPLE = 4096
B, T_MEL, L, V = 32, 1500, 100, 1000


def make_hp(**kw):
    d = dict(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5, encoder_dropout=0.0,
             encoder_type=None, decoder_type='Attention', num_classes=V, eos_id=1, T_norm=True, B_norm=False,
             beam_width=1, max_decoder_seq_len=200, score_func='log_softmax')
    d.update(kw)
    d.setdefault('num_hidden_nodes_encoder', d['num_hidden_nodes'])
    d.setdefault('num_hidden_nodes_decoder', d['num_hidden_nodes'])
    return types.SimpleNamespace(**d)


def make_inputs(hp):
    """The synthetic config-2 batch of SURVEY.md section 8d (what bench.py times): mel (B, 1500, 80) ~ N(0, 1) with
    the padded tail zeroed, mel lengths linspace(1500, 900), targets ~ U{0..V-1} with pad id 0 past
    linspace(100, 50).  Returns mel, mel_lengths, targets, text_lengths (CPU tensors)."""
    g = torch.Generator().manual_seed(SEED)
    mel_len = torch.linspace(T_MEL, 900, B).long()
    text_len = torch.linspace(L, 50, B).long()
    mel = torch.randn(B, T_MEL, hp.lmfb_dim, generator=g)
    tgt = torch.randint(0, hp.num_classes, (B, L), generator=g)
    for b in range(B):
        mel[b, mel_len[b]:] = 0
        tgt[b, text_len[b]:] = 0
    return mel, mel_len, tgt, text_len


def peaked_scale(name):
    """Factor applied to parameter `name` in the "peaked" case.  At random initialisation the encoder's outputs are
    nearly constant over time and the alignment nearly uniform, so the attention's own gradients are ~1e-8 of the
    model's largest and carry no digits in fp32; with larger encoder weights (frames differ), a 10x energy
    projection (peaked softmax over the 500 frames) and 4x context projections (the context matters to the output)
    every attention gradient is within a few 1e-3 of the largest one -- the regime a trained model is in."""
    if 'weight_ih' in name:
        return 3.0
    if 'weight_hh' in name:
        return 2.0
    if name.endswith('att.L_ee.weight'):
        return 10.0
    if '.att.' in name and name.endswith('weight'):
        return 3.0
    if name in ('decoder.L_gy.weight', 'decoder.L_gs.weight'):
        return 4.0
    return 1.0


def scaled_state(state, case):
    if case == "init":
        return {k: v.clone() for k, v in state.items()}
    assert case == "peaked"
    return {k: v * peaked_scale(k) for k, v in state.items()}


def _dirs(numel, seed, n):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, numel, generator=g, dtype=torch.float32).double()


def tensor_digest(t, seed, nproj=NPROJ):
    """[L2 norm, max |.|, nproj projections] of a tensor, accumulated in float64 on the CPU."""
    v = t.detach().reshape(-1).double().cpu()
    out = [float(v.norm()), float(v.abs().max())]
    out += [float(x) for x in (_dirs(v.numel(), seed, nproj) @ v)]
    return np.array(out, dtype=np.float64)


def sample_index(numel, seed=SEED + 1):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (NSAMPLE,), generator=g)


def step_digest(names, params, grads, y, loss, dx):
    """names: parameter names in named_parameters() order; params / grads: the tensors."""
    out = {"loss": np.array(float(loss)), "names": np.array("\n".join(names))}
    yv = y.detach().reshape(-1).double().cpu()
    out["y.sample"] = yv[sample_index(yv.numel())].numpy()
    out["y.stat"] = np.array([float(yv.norm()), float(yv.abs().max())])
    for i, (p, g) in enumerate(zip(params, grads)):
        pv = p.detach().reshape(-1).double().cpu()
        out[f"param.{i}"] = np.array([float(pv.sum()), float(pv.norm())])
        out[f"grad.{i}"] = tensor_digest(g, 1000 + i)
    out["dx"] = tensor_digest(dx, 999, nproj=4)
    return out
