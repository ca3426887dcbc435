"""One BLSTM layer's recurrence kernels (config-2 shape) timed with the saved activations COLD: a 1 GiB fill between
launches pushes G / Cs / dH out of the 256 MiB Infinity Cache, as a training step does (each layer has its own 164 MB
of gates), against the same launches repeated on a warm working set."""
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
junk = torch.empty(256 * 1024 * 1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
def fwd():
    lib.asr_blstm_fwd(W.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H,
                      lengths.data_ptr(), T, B, H, ws.data_ptr(), st)
def bwd():
    lib.asr_blstm_bwd(W.data_ptr(), WT.data_ptr(), G.data_ptr(), Cs.data_ptr(), dH.data_ptr(), B * 2 * H, 2 * H,
                      lengths.data_ptr(), dc.data_ptr(), T, B, H, ws.data_ptr(), st)
TOUCH = os.environ.get("LSTM_TOUCH", "0") == "1"   # after the fill, READ the saved activations once (pulls them back into the Infinity Cache)
def timed(fn, cold, n=5):
    tot = 0.0
    for _ in range(n):
        if cold:
            junk.fill_(1.0)
            if TOUCH:
                sink = G.sum() + Cs.sum() + dH.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
G.copy_(G0); fwd(); bwd(); torch.cuda.synchronize()
# LSTM_DBG: comma-separated asr_debug_set words (bits 12..15 / 16..19: thresholds of the pace rule, 20..27: fixed pace + 1)
for word in [int(x, 0) for x in os.environ.get("LSTM_DBG", "0").split(",")]:
    lib.asr_debug_set(word)
    for cold in (False, True):
        G.copy_(G0)
        tf = timed(fwd, cold)
        tb = timed(bwd, cold)
        print(f"dbg {word:#x} {'cold' if cold else 'warm'}: fwd {tf:7.1f} us ({tf / T:.2f} per step)   bwd {tb:7.1f} us ({tb / T:.2f} per step)", flush=True)
lib.asr_debug_set(0)
print("err", int(ws.view(torch.int32)[1024]))


