"""Recurrence time of one layer at config-2 shape: one launch vs chunked launches (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
T, B, H = 500, 32, 320
g = torch.Generator().manual_seed(0)
W = (torch.randn(2, 4 * H, H, generator=g) * 0.05).to(dev)
WT = W.transpose(1, 2).contiguous()
G0 = torch.randn(T, B, 2, 4 * H, generator=g).to(dev)
G = G0.clone()
Cs = torch.empty(T, B, 2, H, device=dev)
Hout = torch.empty(T, B, 2 * H, device=dev)
dH = torch.randn(T, B, 2 * H, generator=g).to(dev)
lengths = torch.linspace(500, 300, B).to(torch.int32).to(dev)
ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev)
dc = torch.empty(2 * B * H, device=dev)
st = torch.cuda.current_stream().cuda_stream


def chunks(n):
    tc = -(-T // n)
    return [(s, min(T, s + tc)) for s in range(0, T, tc)]


def fwd(n):
    for s0, s1 in chunks(n):
        lib.asr_blstm_fwd_range(W.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H,
                                lengths.data_ptr(), T, B, H, s0, s1, ws.data_ptr(), st)


def bwd(n):
    for s0, s1 in chunks(n):
        lib.asr_blstm_bwd_range(W.data_ptr(), WT.data_ptr(), G.data_ptr(), Cs.data_ptr(), dH.data_ptr(), B * 2 * H,
                                2 * H, lengths.data_ptr(), dc.data_ptr(), T, B, H, s0, s1, ws.data_ptr(), st)


def timeit(f, n, reps=5):
    f(n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f(n)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (1, 2, 4, 8, 1):
    G.copy_(G0)
    tf = timeit(fwd, n)
    tb = timeit(bwd, n)
    print(f"chunks={n}: fwd {tf:8.1f} us  bwd {tb:8.1f} us", flush=True)
