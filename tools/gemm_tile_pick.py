import os, sys
sys.path.insert(0, "/root/repo")
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
for (M, N, K) in ((12795, 640, 2560), (12795, 2560, 640), (11968, 512, 256), (11968, 768, 256), (9600, 640, 2560)):
    A, Bm, Cm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    dt = time_kernel_events(lambda: ops.matmul_nn(A, Bm, Cm), 10)
    print(os.environ.get("ASR_GEMM_BF_TILE"), M, N, K, f"{dt*1e6:.1f} us {2.0*M*N*K/dt/1e12:.1f} TF", flush=True)
