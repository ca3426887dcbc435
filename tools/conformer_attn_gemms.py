"""The per-(utterance, head) products of the relative-position attention at config 5 (B 32, H 4, T 374, d_k 64),
each as the single asr_sgemm launch the encoder issues: time and rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
from asr_amd.Models.Transformer_encoder import _head_gemm, _pad4
from bench import time_kernel_events
dev = torch.device("cuda:0")
B, H, T, dk = 32, 4, 374, 64
d, Tp = H * dk, _pad4(T)
proj, score, posp = (d, dk, T * d), (Tp, T * Tp, H * T * Tp), (d, dk, 0)
q = torch.randn(B * T, d, device=dev); k = torch.randn(B * T, d, device=dev); pp = torch.randn(T, d, device=dev)
att = torch.randn(B, H, T, Tp, device=dev); out = torch.empty(B * T, d, device=dev); dpp = torch.zeros(T, d, device=dev)
fl = 2.0 * B * H * T * T * dk
cases = {
    "ac = (q+u) k^T        (nt)": lambda: _head_gemm(q, k, att, T, T, dk, H, B, a=proj, b=proj, c=score),
    "bd = (q+v) p^T        (nt)": lambda: _head_gemm(q, pp, att, T, T, dk, H, B, a=proj, b=posp, c=score),
    "ctx = attn v          (nn)": lambda: _head_gemm(att, k, out, T, dk, T, H, B, a=score, b=proj, c=proj, b_mode=1),
    "dv = attn^T dctx      (tn)": lambda: _head_gemm(att, q, out, T, dk, T, H, B, a=score, b=proj, c=proj, a_mode=1, b_mode=1, pad_ok=1),
    "dp = sum_b dbd^T (q+v)(tn)": lambda: ops.sgemm(att, q, dpp, T, dk, B * T, a_mode=1, b_mode=1, a_two=(T, Tp, H * T * Tp),
                                                    b_stride=d, c_stride=d, nbatch=H, batch_a=T * Tp, batch_b=dk, batch_c=dk, pad_ok=1),
}
for sk in (4, 8, 16):
    cases[f"dp, splitk {sk}"] = (lambda sk=sk: (dpp.zero_(), ops.sgemm(att, q, dpp, T, dk, B * T, a_mode=1, b_mode=1,
                                 a_two=(T, Tp, H * T * Tp), b_stride=d, c_stride=d, nbatch=H, batch_a=T * Tp, batch_b=dk,
                                 batch_c=dk, pad_ok=1, splitk=sk)))
for name, fn in cases.items():
    dt = time_kernel_events(fn, 10)
    print(f"{name:30s} {dt*1e6:7.1f} us  {fl/dt/1e12:6.1f} TFLOP/s", flush=True)
