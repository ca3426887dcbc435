"""GEMM time per tile shape on the step's own products (ASR_GEMM_TILE is read once per process:
run once per value).  Shapes: the ragged config-2 batch, 12 800 listed frames of 16 000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from bench import time_kernel_events
dev = torch.device("cuda:0")
R = 16000
H = 320
for n in (12800, 16000, 14400, 11200):
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    res = []
    for name, D in (("proj640", 640), ("proj240", 240)):
        X = torch.randn(R, D, device=dev); W = torch.randn(4 * H, D, device=dev); G = torch.empty(R, 8 * H, device=dev)
        dt = time_kernel_events(lambda: ops.sgemm(X, W, G, n, 4 * H, D, a_stride=D, b_stride=D, c_stride=8 * H, m_idx=idx), 10)
        res.append((name, dt, 2.0 * n * 4 * H * D))
    dG = torch.randn(R, 8 * H, device=dev); W = torch.randn(4 * H, 640, device=dev); dX = torch.empty(R, 640, device=dev)
    dt = time_kernel_events(lambda: ops.sgemm(dG, W, dX, n, 640, 4 * H, a_stride=8 * H, b_mode=1, b_stride=640, c_stride=640, m_idx=idx), 10)
    res.append(("dX640", dt, 2.0 * n * 4 * H * 640))
    for name, D in (("wgrad640", 640), ("wgrad320", 320), ("wgrad240", 240)):
        X = torch.randn(R, D, device=dev); gW = torch.zeros(4 * H, D, device=dev)
        sk = ops.pick_splitk(4 * H, D, n)
        dt = time_kernel_events(lambda: ops.sgemm(dG, X, gW, 4 * H, D, n, a_mode=1, a_stride=8 * H, b_mode=1, b_stride=D, c_stride=D, splitk=sk, k_idx=idx), 10)
        res.append((f"{name}/sk{sk}", dt, 2.0 * n * 4 * H * D))
    print(f"tile={os.environ.get('ASR_GEMM_TILE', 'default')} rows={n}: " +
          "  ".join(f"{nm} {dt*1e6:6.1f}us {fl/dt/1e12:5.1f}TF" for nm, dt, fl in res), flush=True)
