"""asr_sgemm in its three arithmetic modes (f32 / bf16x3 / bf16) on the config-2 shapes: error against a float64
product (max |err| / max |want|) and time per launch (HIP events).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib, ops
from bench import time_kernel_events

lib = _lib.load()
dev = torch.device("cuda:0")
T, B, H = 400, 32, 320
M = T * B
g = torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)
X = rnd(M, 2 * H); W = rnd(4 * H, 2 * H) * 0.05; G = rnd(M, 8 * H); Hh = rnd(M, 2 * H)
W8 = rnd(8 * H, 2 * H) * 0.05
cases = []
C1 = torch.empty(M, 8 * H, device=dev)
cases.append(("NT proj  12800x2560x640", 2.0 * M * 8 * H * 2 * H, C1,
              lambda: ops.sgemm(X, W8, C1, M, 8 * H, 2 * H, a_stride=2 * H, b_stride=2 * H, c_stride=8 * H),
              lambda: X.double() @ W8.double().t()))
C2 = torch.empty(M, 2 * H, device=dev)
cases.append(("NN dX    12800x640x2560", 2.0 * M * 8 * H * 2 * H, C2,
              lambda: ops.sgemm(G, W8, C2, M, 2 * H, 8 * H, a_stride=8 * H, b_mode=1, b_stride=2 * H, c_stride=2 * H),
              lambda: G.double() @ W8.double()))
C3 = torch.zeros(4 * H, 2 * H, device=dev)
sk = ops.pick_splitk(4 * H, 2 * H, M)
def dwih():
    C3.zero_()
    ops.sgemm(G, X, C3, 4 * H, 2 * H, M, a_mode=1, a_stride=8 * H, b_mode=1, b_stride=2 * H, c_stride=2 * H, splitk=sk)
cases.append((f"TN dWih  1280x640x12800 sk{sk}", 2.0 * M * 4 * H * 2 * H, C3, dwih,
              lambda: G[:, :4 * H].double().t() @ X.double()))
C4 = torch.zeros(4 * H, H, device=dev)
sk4 = ops.pick_splitk(4 * H, H, M)
def dwhh():
    C4.zero_()
    ops.sgemm(G, Hh, C4, 4 * H, H, M, a_mode=1, a_stride=8 * H, b_mode=1, b_stride=2 * H, c_stride=H, splitk=sk4)
cases.append((f"TN dWhh  1280x320x12800 sk{sk4}", 2.0 * M * 4 * H * H, C4, dwhh,
              lambda: G[:, :4 * H].double().t() @ Hh[:, :H].double()))
U = rnd(3200, 320); Wy = rnd(1000, 320) * 0.05; Y = torch.empty(3200, 1000, device=dev)
cases.append(("NT L_yy  3200x1000x320", 2.0 * 3200 * 1000 * 320, Y,
              lambda: ops.sgemm(U, Wy, Y, 3200, 1000, 320, a_stride=320, b_stride=320, c_stride=1000),
              lambda: U.double() @ Wy.double().t()))
for name, flops, out, fn, want_fn in cases:
    want = want_fn().cpu()
    line = f"{name:32s}"
    for mode, tag in ((0, "f32"), (1, "bf16x3"), (2, "bf16")):
        lib.asr_gemm_precision(mode)
        fn()
        torch.cuda.synchronize()
        err = float((out.cpu().double() - want).abs().max() / want.abs().max())
        dt = time_kernel_events(fn, 10)
        line += f" | {tag}: {dt*1e6:8.1f} us {flops/dt/1e12:6.1f} TF err {err:.1e}"
    print(line, flush=True)
lib.asr_gemm_precision(1)
