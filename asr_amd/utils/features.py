"""Device-side input pipeline -- the host half.

The reference prepares every utterance on the host, one at a time, in its Dataset
(datasets.py:110-120): normalise, SpecAugment, then collate_fn pads (datasets.py:225-247) and
train.py stacks frames (utils/utils.py:47-65).  Here the raw padded batch goes to the GPU once
and `featurize` does all of it in three launches (asr_featurize).  The random mask positions
stay on the host: `draw_spec_aug` replays the reference's own sequence of `random.randrange`
calls (datasets.py:52-81,113-118), so a seeded `random.Random` gives the reference's masks.
"""
import ctypes as C
import random as _random

import torch

from .. import _lib
from .._lib import check
from .utils import lengths_to_device


def draw_spec_aug(mel_length, hp, rng=_random, num_channels=None):
    """Mask positions of ONE utterance, drawn as TrainDatasets.__getitem__ draws them.
    Returns (freq, time): lists of [begin, end) pairs (possibly empty ranges).

    Quirks kept: a zero-width draw ends that mask loop (`return cloned`, datasets.py:59,75);
    the end of a mask is itself drawn (`mask_end = randrange(start, start + width)`), so a mask
    is usually narrower than its width draw; T = min(len // 2 - 1, hp.max_width_T)."""
    C_ = hp.lmfb_dim if num_channels is None else num_channels
    gran = getattr(hp, "granularity", 1)
    freq, time = [], []
    for _ in range(hp.num_mask_F):                       # _freq_mask, datasets.py:52-63
        f = rng.randrange(0, hp.max_width_F, gran)
        f_zero = rng.randrange(0, C_ - f)
        if f * gran == 0:
            break
        mask_end = rng.randrange(f_zero, f_zero + f, gran)
        freq.append((f_zero, mask_end))
    T = min(mel_length // 2 - 1, hp.max_width_T)         # datasets.py:116
    for _ in range(hp.num_mask_T):                       # _time_mask, datasets.py:65-81
        t = rng.randrange(0, T)
        t_zero = rng.randrange(0, mel_length - t)
        if t == 0:
            break
        mask_end = rng.randrange(t_zero, t_zero + t)
        time.append((t_zero, mask_end))
    return freq, time


def masks_to_device(freq_masks, time_masks, device):
    """Per-utterance lists of [begin,end) pairs -> ((B,nF,2), (B,nT,2)) int32 device tensors (None
    for an empty kind), staged through ONE pinned buffer and copied asynchronously."""
    def pack(lists):
        if lists is None:
            return [], 0
        n = max((len(m) for m in lists), default=0)
        flat = []
        for ranges in lists:
            for a0, a1 in ranges:
                flat += (int(a0), int(a1))
            flat += [0, 0] * (n - len(ranges))          # empty ranges pad the shorter lists
        return flat, n
    ff, nF = pack(freq_masks)
    tf, nT = pack(time_masks)
    if not ff and not tf:
        return None, None
    staged = torch.empty(len(ff) + len(tf), dtype=torch.int32, pin_memory=True)
    staged.copy_(torch.tensor(ff + tf, dtype=torch.int32))
    both = staged.to(device, non_blocking=True)
    fm = both[:len(ff)].view(-1, nF, 2) if nF else None
    tm = both[len(ff):].view(-1, nT, 2) if nT else None
    return fm, tm


def featurize(raw, mel_lengths, stack, mean=None, var=None, freq_masks=None, time_masks=None, fill_zero=False):
    """raw (B,Tm,F) fp32 on the GPU (padded batch of unnormalised features), mel_lengths (B,).
    freq_masks / time_masks: per-utterance lists of [begin,end) pairs (from draw_spec_aug), the
    device tensors masks_to_device made of them, or None.
    Returns (x (B, Tm // stack, F * stack), mel_lengths // stack) -- what train.py:48-49 hands to
    the model."""
    if not raw.is_cuda:
        raise RuntimeError("asr_amd.featurize: the batch must be on the GPU; no CPU fallback exists")
    lib = _lib.load()
    raw = raw.float().contiguous()
    B, Tm, F = raw.shape
    dev = raw.device
    if torch.is_tensor(freq_masks) or torch.is_tensor(time_masks):      # already on the device
        fm, tm = freq_masks, time_masks
    else:
        fm, tm = masks_to_device(freq_masks, time_masks, dev)
    nF = 0 if fm is None else fm.shape[1]
    nT = 0 if tm is None else tm.shape[1]
    len_dev = lengths_to_device(mel_lengths, dev)
    if (mean is None) != (var is None):
        raise RuntimeError("asr_amd.featurize: mean and var come together (hp.mean_file and hp.var_file)")
    if mean is not None:
        mean = mean.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        var = var.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if mean.numel() != F or var.numel() != F:
            raise RuntimeError("asr_amd.featurize: mean/var must have one value per channel")
    ws = torch.empty(int(lib.asr_featurize_ws_floats(B, Tm, F, nF)), device=dev, dtype=torch.float32)
    out = torch.empty(B, Tm // stack, F * stack, device=dev, dtype=torch.float32)
    p = lambda t: 0 if t is None else t.data_ptr()
    check(lib.asr_featurize(raw.data_ptr(), len_dev.data_ptr(), p(mean), p(var), p(fm), nF, int(bool(fill_zero)),
                            p(tm), nT, int(stack), B, Tm, F, ws.data_ptr(), out.data_ptr(),
                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out, mel_lengths // stack
