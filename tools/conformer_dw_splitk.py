"""Weight-gradient GEMMs of the Conformer's d_model-sized layers (out (256 x 256) = dy^T x, K = 11968 rows):
rate vs split-K factor (ops.pick_splitk's choice first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
for M, N, K in ((256, 256, 11968), (512, 256, 11968), (2048, 256, 11968), (256, 2048, 11968)):
    A, Bm, Cm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)
    line = [f"{M}x{N}x{K}: pick={ops.pick_splitk(M, N, K)}"]
    for sk in (ops.pick_splitk(M, N, K), 4, 8, 16, 24, 32, 48, 64):
        def fn():
            if sk > 1:
                Cm.zero_()
            ops.sgemm(A, Bm, Cm, M, N, K, a_mode=1, a_stride=M, b_mode=1, b_stride=N, c_stride=N, splitk=sk)
        dt = time_kernel_events(fn, 10)
        line.append(f"sk{sk}: {dt*1e6:.0f}us {2.0*M*N*K/dt/1e12:.0f}T")
    print("  ".join(line), flush=True)
