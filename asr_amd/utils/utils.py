"""Host-side helpers the training loop calls around the hot path."""


def frame_stacking(x, x_lengths, stack):
    """Mirror of utils/utils.py:47-65: drop the tail frames, fold `stack` frames
    into the feature axis (a view, no kernel), divide the lengths."""
    if stack == 1:
        return x, x_lengths
    n = x.shape[1] // stack
    return x[:, : n * stack].reshape(x.shape[0], n, -1), x_lengths // stack


def lengths_to_device(lengths, device):
    """(B,) lengths -> int32 tensor on `device` WITHOUT stalling the host.

    The reference hands `lengths` over on the CPU (pack_padded_sequence wants them there,
    Models/encoder.py:32).  A plain `.to(device)` of a pageable CPU tensor is a synchronous
    hipMemcpy: stream-ordered behind everything already enqueued, so the host would block until
    the previous step's GPU work has drained and then start enqueuing the next step into an idle
    GPU (measured: ~1.5 ms of launch gaps per step).  Staging through a pinned buffer makes the
    copy asynchronous; the caching host allocator keeps the buffer alive until it has run.
    A tensor already on the device is only converted."""
    import torch
    if lengths.is_cuda:
        return lengths.to(device=device, dtype=torch.int32)
    staged = torch.empty(lengths.shape, dtype=torch.int32, pin_memory=True)
    staged.copy_(lengths)
    return staged.to(device, non_blocking=True)
