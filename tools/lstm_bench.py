"""Times one BLSTM layer's recurrence kernels (config-2 shape) with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from bench import time_kernel_events
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
def fwd():
    lib.asr_blstm_fwd(W.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H,
                      lengths.data_ptr(), T, B, H, ws.data_ptr(), st)
def bwd():
    lib.asr_blstm_bwd(W.data_ptr(), WT.data_ptr(), G.data_ptr(), Cs.data_ptr(), dH.data_ptr(), B * 2 * H, 2 * H,
                      lengths.data_ptr(), dc.data_ptr(), T, B, H, ws.data_ptr(), st)
for prec in (0, 1):
    lib.asr_blstm_precision(prec)
    for dbg in (0, 1, 16):
        lib.asr_debug_set(dbg)
        G.copy_(G0)
        tf = time_kernel_events(fwd, 3)
        tb = time_kernel_events(bwd, 3)
        print(f"precision={prec} dbg={dbg}: fwd {tf*1e6/T:6.2f} us/step   bwd {tb*1e6/T:6.2f} us/step", flush=True)
lib.asr_debug_set(0)
print("err", int(ws.view(torch.int32)[1024]))
