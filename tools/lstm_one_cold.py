"""One forward + one backward BLSTM layer recurrence at the config-2 shape with the saved activations COLD (a 1 GiB
fill before every launch, as tools/lstm_cold.py) -- for rocprofv3 --pmc passes; ASR_BLSTM_PREFETCH=0 / 16 switches
the prefetching workgroups of the XCD-grouped grid off / on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
T, B, H = 500, 32, 320
g = torch.Generator().manual_seed(0)
W = (torch.randn(2, 4 * H, H, generator=g) * 0.05).to(dev)
G = torch.randn(T, B, 2, 4 * H, generator=g).to(dev)
Cs = torch.empty(T, B, 2, H, device=dev)
Hout = torch.empty(T, B, 2 * H, device=dev)
dH = torch.randn(T, B, 2 * H, generator=g).to(dev)
lengths = torch.linspace(500, 300, B).to(torch.int32).to(dev)
ws = torch.zeros(int(lib.asr_blstm_ws_floats(B, H)), device=dev)
dc = torch.empty(2 * B * H, device=dev)
junk = torch.empty(256 * 1024 * 1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    junk.fill_(1.0)
    lib.asr_blstm_fwd(W.data_ptr(), G.data_ptr(), Cs.data_ptr(), Hout.data_ptr(), B * 2 * H, 2 * H,
                      lengths.data_ptr(), T, B, H, ws.data_ptr(), st)
    junk.fill_(2.0)
    lib.asr_blstm_bwd(W.data_ptr(), W.data_ptr(), G.data_ptr(), Cs.data_ptr(), dH.data_ptr(), B * 2 * H, 2 * H,
                      lengths.data_ptr(), dc.data_ptr(), T, B, H, ws.data_ptr(), st)
torch.cuda.synchronize()
assert int(ws.view(torch.int32)[1024]) == 0
