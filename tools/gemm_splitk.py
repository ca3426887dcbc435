"""Weight-gradient products (A^T B, K = all frames) by split-K factor: microseconds per launch (C zeroed outside the
timed region; the split-K partials are summed with float atomics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
shapes = [("c5 ff dW1", 2048, 256, 11968), ("c5 ff dW2", 256, 2048, 11968), ("c5 d256 dW", 256, 256, 11968),
          ("c2 dW_ih", 2560, 640, 16000), ("c2 dW_hh", 1280, 320, 16000), ("c2 dW_ih layer0", 2560, 240, 16000),
          ("c5 conv2 dW", 256, 2304, 227392)]
for name, M, N, K in shapes:
    A, Bm, Cm = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)
    line = f"{name:16s} M={M:5d} N={N:5d} K={K:6d} default sk={ops.pick_splitk(M, N, K):3d}:"
    for sk in [int(x) for x in os.environ.get('SKS', '1,2,4,8,12,16,24,32,48,64,96,128').split(',')]:
        if K // sk < 64:
            continue
        fn = lambda: ops.sgemm(A, Bm, Cm, M, N, K, a_mode=1, a_stride=M, b_mode=1, b_stride=N, c_stride=N, splitk=sk)
        dt = time_kernel_events(fn, 10)
        line += f"  {sk}:{dt * 1e6:.0f}"
    print(line, flush=True)
    del A, Bm, Cm
