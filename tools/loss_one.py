"""The fused label-smoothed CE once at the config-4 (CSJ, 7k BPE) shape: B=32, L=150, V=7000 (for rocprofv3 --pmc)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
B, L, V = 32, int(os.environ.get("LOSS_L", "150")), int(os.environ.get("LOSS_V", "7000"))
g = torch.Generator().manual_seed(0)
z = torch.randn(B, L, V, generator=g).to(dev)
tg = torch.randint(0, V, (B, L), generator=g).to(dev)
tl = torch.full((B,), L, dtype=torch.int64, device=dev)
row = torch.empty(B * L, device=dev)
ls = torch.empty((), device=dev)
dz = torch.empty_like(z)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.asr_ls_ce_fwd_bwd(z.data_ptr(), tg.data_ptr(), tl.data_ptr(), B, L, V, C.c_float(0.1), 1, 0,
                          row.data_ptr(), ls.data_ptr(), dz.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.asr_ls_ce_fwd_bwd(z.data_ptr(), tg.data_ptr(), tl.data_ptr(), B, L, V, C.c_float(0.1), 1, 0,
                          row.data_ptr(), ls.data_ptr(), dz.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"ls_ce B={B} L={L} V={V}: {us:.1f} us per call, {2.0 * B * L * V * 4 / us / 1e3:.0f} GB/s of algorithmic bytes (logits read + gradient written)")
