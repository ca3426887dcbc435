"""Which bf16x3 tile the N = 256 / 512 / 768 products of the Conformer step should take: the same product with the tile
forced by ASR_GEMM_BF_TILE (1 = 128 x 128, 2 = 128 x 64, 3 = 64 x 64), fresh library load per setting.
usage: for t in 1 2 3; do ASR_GEMM_BF_TILE=$t python tools/gemm_tile_pick.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
for (M, N, K, mode) in ((11968, 256, 256, "nt"), (11968, 256, 2048, "nt"), (11968, 256, 256, "nn"), (11968, 256, 2048, "nn"),
                        (11968, 2048, 256, "nt"), (11968, 512, 256, "nt"), (11968, 768, 256, "nt")):
    A, Cm = torch.randn(M, K, device=dev), torch.empty(M, N, device=dev)
    if mode == "nt":
        W = torch.randn(N, K, device=dev)
        fn = lambda: ops.linear_nt(A, W, Cm)
    else:
        W = torch.randn(K, N, device=dev)
        fn = lambda: ops.matmul_nn(A, W, Cm)
    dt = time_kernel_events(fn, 20)
    print("tile", os.environ.get("ASR_GEMM_BF_TILE", "auto"), mode, M, N, K, f"{dt*1e6:7.1f} us {2.0*M*N*K/dt/1e12:6.1f} TF", flush=True)
