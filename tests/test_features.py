"""Input pipeline.  Parity unpinned by the reference (datasets.py does not
import here, see oracle/ref_features.py): the tests hold the HIP path and the host-side mask
draws to the line-by-line restatement."""
import random

import pytest
import torch

from oracle import ref_features as RF
from oracle import ref_model as R


def _hp(**kw):
    d = dict(lmfb_dim=8, frame_stacking=3, num_mask_F=2, num_mask_T=2, max_width_F=5, max_width_T=20, granularity=1)
    d.update(kw)
    return R.make_hp(**d)


@pytest.mark.parametrize("seed", range(6))
def test_host_draws_replay_the_reference_sequence(seed):
    """draw_spec_aug consumes the random stream exactly like _freq_mask/_time_mask do, and the
    positions it returns reproduce the oracle's masked utterance."""
    from asr_amd.utils.features import draw_spec_aug
    hp = _hp(num_mask_F=3, num_mask_T=3, max_width_F=4 if seed % 2 else 7)
    g = torch.Generator().manual_seed(seed)
    mel = torch.randn(40 + 7 * seed, hp.lmfb_dim, generator=g, dtype=torch.float64)
    want = RF.spec_aug_one(mel, hp, random.Random(seed))
    rng = random.Random(seed)
    freq, time = draw_spec_aug(mel.shape[0], hp, rng)
    probe = random.Random(seed)
    RF.spec_aug_one(mel, hp, probe)
    assert rng.random() == probe.random()            # same number of draws consumed
    x = mel.clone()
    for f0, f1 in freq:
        x[:, f0:f1] = x.mean()
    for t0, t1 in time:
        x[t0:t1] = 0
    assert torch.equal(x, want)


@pytest.mark.gpu
@pytest.mark.parametrize("lens,stack,norm,aug", [([50, 37, 20], 3, True, True), ([64, 64], 1, False, True),
                                                 ([33, 90, 12, 61], 2, True, False), ([130], 3, False, False),
                                                 ([200, 150, 170, 40, 90], 3, True, True)])
def test_featurize_vs_oracle(lens, stack, norm, aug):
    from asr_amd.utils.features import draw_spec_aug, featurize
    assert torch.cuda.is_available(), "GPU tests need a GPU: refusing to pass on a fallback"
    dev = torch.device("cuda:0")
    hp = _hp(frame_stacking=stack, lmfb_dim=12 if stack == 2 else 8)
    F = hp.lmfb_dim
    g = torch.Generator().manual_seed(sum(lens))
    utts = [torch.randn(n, F, generator=g) * 3 + 1 for n in lens]
    mean = torch.randn(F, generator=g) if norm else None
    var = (torch.rand(F, generator=g) + 0.5) if norm else None
    want, want_len = RF.pipeline([u.double() for u in utts], hp, random.Random(7),
                                 None if mean is None else mean.double(), None if var is None else var.double(), aug)
    rng = random.Random(7)
    masks = [draw_spec_aug(n, hp, rng) for n in lens] if aug else None
    Tm = max(lens)
    raw = torch.full((len(lens), Tm, F), 123.0)       # garbage in the padding: must come out as zeros
    for b, u in enumerate(utts):
        raw[b, :lens[b]] = u
    x, xl = featurize(raw.to(dev), torch.tensor(lens), stack, mean, var,
                      None if masks is None else [m[0] for m in masks],
                      None if masks is None else [m[1] for m in masks])
    assert tuple(x.shape) == tuple(want.shape) and torch.equal(xl, want_len)
    err = float((x.cpu().double() - want).abs().max()) / float(want.abs().max())
    assert err < 2e-6
    for b, n in enumerate(lens):                      # padded frames are exactly zero
        flat = x[b].reshape(-1, F)
        assert float(flat[n:].abs().max() if n < flat.shape[0] else 0.0) == 0.0


@pytest.mark.gpu
def test_featurize_errors():
    from asr_amd.utils.features import featurize
    with pytest.raises(RuntimeError):
        featurize(torch.zeros(1, 4, 3), torch.tensor([4]), 1)
    x = torch.zeros(1, 4, 3, device="cuda:0")
    with pytest.raises(RuntimeError):
        featurize(x, torch.tensor([4]), 1, mean=torch.zeros(3))
    with pytest.raises(RuntimeError):
        featurize(x, torch.tensor([4]), 1, mean=torch.zeros(2), var=torch.ones(2))
