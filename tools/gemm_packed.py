"""asr_sgemm with the B operand (weights) pre-split into {hi:lo} words against the same product splitting on the fly:
bit-identical results, time per launch (config-2 projection and dX shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
dev = torch.device("cuda:0")

def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, M, N, K, b_mode in (("projection", 12800, 2560, 640, 0), ("dX", 12800, 640, 2560, 1), ("probe", 16000, 1280, 640, 0)):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) if b_mode == 0 else torch.randn(K, N, device=dev)
    Wp = ops.pack_hilo(W).view_as(W)
    Ap = ops.pack_hilo(A).view_as(A)
    C0, C1, C2 = (torch.empty(M, N, device=dev) for _ in range(3))
    bs = K if b_mode == 0 else N
    f0 = lambda: ops.sgemm(A, W, C0, M, N, K, a_stride=K, b_mode=b_mode, b_stride=bs, c_stride=N)
    f1 = lambda: ops.sgemm(A, Wp, C1, M, N, K, a_stride=K, b_mode=b_mode, b_stride=bs, c_stride=N, b_packed=True)
    t0, t1 = timed(f0), timed(f1)
    msg = f"{name:10s} {M} x {N} x {K} b_mode {b_mode}: fp32 operands {t0:7.1f} us, B pre-split {t1:7.1f} us, equal {torch.equal(C0, C1)}"
    if b_mode == 0:
        f2 = lambda: ops.sgemm(Ap, Wp, C2, M, N, K, a_stride=K, b_mode=0, b_stride=bs, c_stride=N, a_packed=True, b_packed=True)
        t2 = timed(f2)
        msg += f", A and B pre-split {t2:7.1f} us, equal {torch.equal(C0, C2)}"
    print(msg, flush=True)
