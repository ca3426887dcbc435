"""Random shapes against the oracle: tests/random_shapes.py with a fixed seed (encoder, decoder and Conformer encoder
at widths that are and are not multiples of 4 / 16, batches on both sides of the persistent kernels' limit of 32,
lengths down to 1 and 0, a single label step), forward and every gradient, to the bars of test_gpu_parity.py."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 202])
def test_random_shapes_vs_oracle(seed):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU: refusing to pass on a fallback"
    from tests import random_shapes
    assert random_shapes.run(150, seed, quiet=True) == 0
