"""GEMM rates at the Conformer (config 5) shapes, per tile configuration (ASR_GEMM_TILE is read once per process,
so the tool re-runs itself per setting): rows = B*T' = 11968, d = 256, d_ff = 2048, conv2 im2col K = 2304."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "ASR_GEMM_TILE" not in os.environ:
    for t in ("2", "3", "0"):
        subprocess.run([sys.executable, __file__], env=dict(os.environ, ASR_GEMM_TILE=t), check=True)
    sys.exit(0)
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
R = 11968
shapes = [  # name, kind, M, N, K
    ("ff.linear1 fwd", "nt", R, 2048, 256), ("ff.linear2 fwd", "nt", R, 256, 2048), ("qkv/out fwd", "nt", R, 256, 256),
    ("ff.linear1 dx", "nn", R, 256, 2048), ("ff.linear2 dx", "nn", R, 2048, 256), ("d256 dx", "nn", R, 256, 256),
    ("ff dW1", "tn", 2048, 256, R), ("ff dW2", "tn", 256, 2048, R), ("d256 dW", "tn", 256, 256, R),
    ("conv2 fwd", "nt", 227392, 256, 2304), ("conv2 dcol", "nn", 227392, 2304, 256), ("conv2 dW", "tn", 256, 2304, 227392),
]
out = {}
for name, kind, M, N, K in shapes:
    if kind == "nt":
        A, Bm, Cm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: ops.linear_nt(A, Bm, Cm)
    elif kind == "nn":
        A, Bm, Cm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: ops.matmul_nn(A, Bm, Cm)
    else:
        A, Bm, Cm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: ops.matmul_tn(A, Bm, Cm)
    dt = time_kernel_events(fn, 10)
    out[name] = round(2.0 * M * N * K / dt / 1e12, 1)
    del A, Bm, Cm
print("ASR_GEMM_TILE=" + os.environ["ASR_GEMM_TILE"], json.dumps(out), flush=True)
