"""GEMM rate vs M (tail / wave-quantisation check): NT, N=1280, K=640."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
N, K = 1280, 640
for M in (8192, 16000, 16384, 32768, 65536, 131072):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    dt = time_kernel_events(lambda: ops.sgemm(X, W, C, M, N, K, a_stride=K, b_stride=K, c_stride=N), 10)
    print(f"M={M:7d} blocks={((M+127)//128)*(N//64):6d} {dt*1e6:9.1f} us {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
