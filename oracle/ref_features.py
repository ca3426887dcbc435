"""CPU oracle for the input pipeline.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's datasets.py cannot be imported in the build container (it
imports librosa at module level, which is not installed, and the reference holds no fixtures
or tests for this code), so no golden vector pins this restatement.  It follows the source
line by line instead:

* normalisation ......... datasets.py:110-112   (x -= mean; x /= sqrt(var), per channel)
* _freq_mask ............ datasets.py:52-63     (fill = cloned.mean() taken INSIDE the loop)
* _time_mask ............ datasets.py:65-81     (called with replace_with_zero=True, :118)
* the call .............. datasets.py:113-118   (T = min(len // 2 - 1, hp.max_width_T))
* padding ............... datasets.py:225-247   (collate_fn: zeros up to the longest utterance)
* frame stacking ........ utils/utils.py:47-65

`spec_aug_one` draws from the `rng` it is given with the same calls, in the same order, as the
reference draws from the `random` module.
"""
import torch

from .ref_model import frame_stacking


def freq_mask(spec, rng, F=10, num_masks=1, replace_with_zero=False, granularity=1):
    cloned = spec.clone()
    num_mel_channels = cloned.shape[1]
    for _ in range(num_masks):
        f = rng.randrange(0, F, granularity)
        f_zero = rng.randrange(0, num_mel_channels - f)
        if f_zero == f_zero + f * granularity:
            return cloned
        mask_end = rng.randrange(f_zero, f_zero + f, granularity)
        if replace_with_zero:
            cloned[:, f_zero:mask_end] = 0
        else:
            cloned[:, f_zero:mask_end] = cloned.mean()
    return cloned


def time_mask(spec, rng, T=50, num_masks=1, replace_with_zero=False):
    cloned = spec.clone()
    len_spectro = cloned.shape[0]
    for _ in range(num_masks):
        t = rng.randrange(0, T)
        t_zero = rng.randrange(0, len_spectro - t)
        if t_zero == t_zero + t:
            return cloned
        mask_end = rng.randrange(t_zero, t_zero + t)
        if replace_with_zero:
            cloned[t_zero:mask_end, :] = 0
        else:
            cloned[t_zero:mask_end, :] = cloned.mean()
    return cloned


def spec_aug_one(mel, hp, rng):
    """datasets.py:113-118 for one (T,F) utterance."""
    T = min(mel.shape[0] // 2 - 1, hp.max_width_T)
    return time_mask(freq_mask(mel, rng, F=hp.max_width_F, num_masks=hp.num_mask_F, granularity=hp.granularity),
                     rng, T=T, num_masks=hp.num_mask_T, replace_with_zero=True)


def pipeline(utterances, hp, rng, mean=None, var=None, use_spec_aug=True):
    """List of raw (T_b,F) utterances -> (x (B,T/stack,F*stack), lengths // stack)."""
    done = []
    for u in utterances:
        x = u.clone()
        if mean is not None:
            x = (x - mean.reshape(1, -1)) / torch.sqrt(var.reshape(1, -1))
        if use_spec_aug:
            x = spec_aug_one(x, hp, rng)
        done.append(x)
    Tm = max(x.shape[0] for x in done)
    batch = torch.zeros(len(done), Tm, done[0].shape[1], dtype=done[0].dtype)
    for b, x in enumerate(done):
        batch[b, :x.shape[0]] = x
    lengths = torch.tensor([x.shape[0] for x in done])
    return frame_stacking(batch, lengths, hp.frame_stacking)
