"""Times the GEMM shapes of the config-2 step with HIP events (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events

dev = torch.device("cuda:0")
T, B, H = 500, 32, 320
M = T * B
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
X = rnd(M, 2 * H); W = rnd(4 * H, 2 * H); G = rnd(M, 8 * H); Hh = rnd(M, 2 * H)
out = {}
def rep(name, flops, fn, it=10):
    dt = time_kernel_events(fn, it)
    print(f"{name:28s} {dt*1e6:9.1f} us  {flops/dt/1e12:7.1f} TFLOP/s", flush=True)
C1 = torch.empty(M, 8 * H, device=dev)
rep("NT proj 16000x1280x640", 2.0 * M * 4 * H * 2 * H,
    lambda: ops.sgemm(X, W, C1, M, 4 * H, 2 * H, a_stride=2 * H, b_stride=2 * H, c_stride=8 * H))
C2 = torch.empty(M, 2 * H, device=dev)
rep("NN dX 16000x640x1280", 2.0 * M * 4 * H * 2 * H,
    lambda: ops.sgemm(G, W, C2, M, 2 * H, 4 * H, a_stride=8 * H, b_mode=1, b_stride=2 * H, c_stride=2 * H))
C3 = torch.zeros(4 * H, 2 * H, device=dev)
sk = ops.pick_splitk(4 * H, 2 * H, M)
rep(f"TN dWih 1280x640x16000 sk{sk}", 2.0 * M * 4 * H * 2 * H,
    lambda: ops.sgemm(G, X, C3, 4 * H, 2 * H, M, a_mode=1, a_stride=8 * H, b_mode=1, b_stride=2 * H,
                      c_stride=2 * H, splitk=sk))
C4 = torch.zeros(4 * H, H, device=dev)
sk = ops.pick_splitk(4 * H, H, M)
rep(f"TN dWhh 1280x320x16000 sk{sk}", 2.0 * M * 4 * H * H,
    lambda: ops.sgemm(G, Hh, C4, 4 * H, H, M, a_mode=1, a_stride=8 * H, b_mode=1, b_stride=2 * H,
                      c_stride=H, splitk=sk))
for s in (4, 8, 16):
    rep(f"TN dWih sk{s}", 2.0 * M * 4 * H * 2 * H,
        lambda: ops.sgemm(G, X, C3, 4 * H, 2 * H, M, a_mode=1, a_stride=8 * H, b_mode=1, b_stride=2 * H,
                          c_stride=2 * H, splitk=s))
