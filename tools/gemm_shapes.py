"""asr_sgemm at the shapes of the training steps (config 2 and config 5), in the library's default arithmetic (bf16x3):
microseconds per launch and fp32-equivalent TFLOP/s (2 M N K / t).  ASR_GEMM_BF_TILE=1|2|3 forces a tile shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
R2, R5 = 16000, 11968
shapes = [  # name, kind, M, N, K
    ("c2 proj fwd (both dirs)", "nt", R2, 2560, 640), ("c2 proj dx", "nn", R2, 640, 2560), ("c2 dW_ih", "tn", 2560, 640, R2),
    ("c2 dW_hh (one dir)", "tn", 1280, 320, R2), ("c2 proj fwd layer 0", "nt", R2, 2560, 240),
    ("c5 ff.linear1 fwd", "nt", R5, 2048, 256), ("c5 ff.linear2 fwd", "nt", R5, 256, 2048), ("c5 d256 fwd", "nt", R5, 256, 256),
    ("c5 ff.linear1 dx", "nn", R5, 256, 2048), ("c5 ff.linear2 dx", "nn", R5, 2048, 256), ("c5 d256 dx", "nn", R5, 256, 256),
    ("c5 ff dW1", "tn", 2048, 256, R5), ("c5 ff dW2", "tn", 256, 2048, R5), ("c5 d256 dW", "tn", 256, 256, R5),
    ("c5 conv2 dcol", "nn", 227392, 2304, 256), ("c5 conv2 dW", "tn", 256, 2304, 227392),
]
only = os.environ.get("GEMM_ONLY")
tot = 0.0
for name, kind, M, N, K in shapes:
    if only and only not in name:
        continue
    if kind == "nt":
        A, Bm, Cm = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: ops.linear_nt(A, Bm, Cm)
    elif kind == "nn":
        A, Bm, Cm = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
        fn = lambda: ops.matmul_nn(A, Bm, Cm)
    else:
        A, Bm, Cm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)
        fn = lambda: ops.matmul_tn(A, Bm, Cm, zeroed=True)
    dt = time_kernel_events(fn, 10)
    tot += dt
    print(f"{name:26s} {kind} M={M:6d} N={N:5d} K={K:6d} {dt * 1e6:8.1f} us {2.0 * M * N * K / dt / 1e12:7.1f} TFLOP/s", flush=True)
    del A, Bm, Cm
print(f"sum {tot * 1e6:.1f} us")
