"""Host-side helpers the training loop calls around the hot path."""


def frame_stacking(x, x_lengths, stack):
    """Mirror of utils/utils.py:47-65: drop the tail frames, fold `stack` frames
    into the feature axis (a view, no kernel), divide the lengths."""
    if stack == 1:
        return x, x_lengths
    n = x.shape[1] // stack
    return x[:, : n * stack].reshape(x.shape[0], n, -1), x_lengths // stack
