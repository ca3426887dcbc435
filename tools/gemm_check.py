"""Correctness + rate of the big GEMM shapes under the current ASR_GEMM_TILE setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 16000, 1280, 640
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.zeros(M, N, device=dev)
ops.sgemm(X, W, C, M, N, K, a_stride=K, b_stride=K, c_stride=N)
ref = (X.double() @ W.double().t())
print("NT max rel err", float((C.double() - ref).abs().max() / ref.abs().max()))
G = torch.randn(M, 2 * N, device=dev); C3 = torch.zeros(N, K, device=dev)
ops.sgemm(G, X, C3, N, K, M, a_mode=1, a_stride=2 * N, b_mode=1, b_stride=K, c_stride=K, splitk=16)
ref3 = G[:, :N].double().t() @ X.double()
print("TN max rel err", float((C3.double() - ref3).abs().max() / ref3.abs().max()))
for it in (10, 50, 200):
    dt = time_kernel_events(lambda: ops.sgemm(X, W, C, M, N, K, a_stride=K, b_stride=K, c_stride=N), it)
    print(f"{it:4d} launches: {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:6.1f} TFLOP/s", flush=True)
