import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never be silently skipped on a GPU box; on a CPU-only box
    # they are deselected by `-m "not gpu"`.  If somebody runs them without a
    # GPU, fail loudly rather than pass on a fallback.
    pass


# asr_sgemm's arithmetic modes (include/asr_amd.h, asr_gemm_precision) with the agreement each is held to against
# a float64 product, max |err| / max |want|: exact fp32 products; the bf16x3 split (the library default: ~16
# mantissa bits per product); plain bf16 products (the opt-in reduced-precision path, 8 mantissa bits).
GEMM_MODES = [("f32", 2e-6), ("bf16x3", 2e-5), ("bf16", 2e-2)]


@pytest.fixture(params=GEMM_MODES, ids=[m for m, _ in GEMM_MODES])
def gemm_mode(request):
    from asr_amd import ops
    name, tol = request.param
    with ops.gemm_precision(name):
        yield tol, name


@pytest.fixture(autouse=True)
def _default_gemm_precision(request):
    """Every GPU test starts and ends in the library's start-up mode, whatever a test before it did."""
    if "gpu" not in request.keywords:
        yield
        return
    from asr_amd import _lib
    lib = _lib.load()
    start, start_rec = lib.asr_gemm_precision(-1), lib.asr_blstm_precision(-1)
    yield
    lib.asr_gemm_precision(start)
    lib.asr_blstm_precision(start_rec)


@pytest.fixture(params=[0, 1], ids=["recurrence-f32", "recurrence-bf16x3"])
def blstm_mode(request):
    """Arithmetic of the recurrent products inside the persistent BLSTM kernels (asr_blstm_precision)."""
    from asr_amd import _lib
    lib = _lib.load()
    old = lib.asr_blstm_precision(request.param)
    yield request.param
    lib.asr_blstm_precision(old)
