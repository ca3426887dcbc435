import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
dev = torch.device("cuda:0")
M, N, K = 16000, 1280, 640
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C1 = torch.empty(M, 2 * N, device=dev)
for _ in range(3):
    ops.sgemm(X, W, C1, M, N, K, a_stride=K, b_stride=K, c_stride=2 * N)
torch.cuda.synchronize()
