"""GPU: the allocation-free entry point asr_blstm_fwd_noalloc (static device scratch, no workspace
argument) against asr_blstm_fwd, through the C ABI, bit for bit (same kernel, same arithmetic), on
two calls in a row -- the second with a SMALLER input, so whatever the first call left in the shared
scratch (flag words, epoch-tagged exchange slots of a larger grid) must not leak into it -- and the
rejection of shapes beyond the static maximum."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, entry, W, G0, lengths, T, B, H, ws=None):
    dev = W.device
    G = G0.clone()
    Cs = torch.full((T, B, 2, H), 7.0, device=dev)
    Hout = torch.full((T, B, 2 * H), 7.0, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [W.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H, lengths.data_ptr(), T, B, H]
    if ws is not None:
        args.append(ws.data_ptr())
    torch.cuda.synchronize()
    before = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev))
    rc = getattr(lib, entry)(*args, st)
    torch.cuda.synchronize()
    # the call itself takes no device memory from torch's allocator (the library has no other)
    assert (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)) == before
    return rc, G, Cs, Hout


def _err_word(lib, dev):
    """The sticky error word of the static scratch, copied back (4 bytes, device -> host)."""
    p = lib.asr_blstm_fwd_noalloc_err_word()
    assert p, lib.asr_last_error()
    host = C.c_uint32(0xdeadbeef)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.byref(host), C.c_void_p(p), 4, 2) == 0        # hipMemcpyDeviceToHost
    return host.value


def test_blstm_fwd_noalloc_matches_workspace_entry_twice():
    from asr_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    dev = torch.device("cuda:0")
    lib = _lib.load()
    assert lib.asr_blstm_fwd_noalloc_max_floats() * 4 <= 512 * 1024          # "modest": 264.25 KiB
    g = torch.Generator().manual_seed(11)
    for (T, B, H) in ((40, 32, 320), (23, 9, 64)):          # second call smaller in every dimension
        W = (torch.randn(2, 4 * H, H, generator=g) * 0.1).to(dev)
        G0 = torch.randn(T, B, 2, 4 * H, generator=g).to(dev)
        lengths = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32)
        lengths[0] = T
        lengths = lengths.to(dev)
        ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev)
        rc0, Gw, Cw, Hw = _run(lib, "asr_blstm_fwd", W, G0, lengths, T, B, H, ws)
        rc1, Gn, Cn, Hn = _run(lib, "asr_blstm_fwd_noalloc", W, G0, lengths, T, B, H)
        assert rc0 == 0 and rc1 == 0, lib.asr_last_error()
        assert torch.equal(Hn, Hw) and torch.equal(Cn, Cw) and torch.equal(Gn, Gw), (T, B, H)
        assert float(Hn.abs().max()) > 0 and bool(torch.isfinite(Hn).all())
        for b in range(B):                                   # rows past the length are exactly 0
            assert float(Hn[int(lengths[b]):, b].abs().max() if int(lengths[b]) < T else 0.0) == 0.0
        assert int(ws.view(torch.int32)[1024].item()) == 0   # no timed-out wait on the reference run
        assert _err_word(lib, dev) == 0                      # ... nor on the allocation-free one
        # the word folds into a value like a workspace's word 1024 does (asr_poison_if_flag): stays finite here
        val = torch.ones((), device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.asr_poison_if_flag(lib.asr_blstm_fwd_noalloc_err_word(), val.data_ptr(), st) == 0
        assert float(val) == 1.0
    # beyond the static maximum: a clear error, nothing launched
    T, B, H = 4, 33, 64
    W = torch.zeros(2, 4 * H, H, device=dev)
    G0 = torch.zeros(T, B, 2, 4 * H, device=dev)
    lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
    rc, _, _, Hn = _run(lib, "asr_blstm_fwd_noalloc", W, G0, lengths, T, B, H)
    assert rc != 0 and b"static scratch" in lib.asr_last_error()
    assert float((Hn - 7.0).abs().max()) == 0.0
    T, B, H = 4, 2, 516
    W = torch.zeros(2, 4 * H, H, device=dev)
    G0 = torch.zeros(T, B, 2, 4 * H, device=dev)
    lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
    rc, _, _, _ = _run(lib, "asr_blstm_fwd_noalloc", W, G0, lengths, T, B, H)
    assert rc != 0 and b"static scratch" in lib.asr_last_error()
