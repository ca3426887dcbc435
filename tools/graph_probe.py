"""Does a HIP graph shorten the launch-bound decoder loop?  Times the decoder forward (100 label
steps x 4 kernels + hoisted GEMMs) issued kernel by kernel against one replay of the same work
captured in a graph (torch.cuda.CUDAGraph -> hipGraphLaunch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_model as R
from asr_amd.Models.AttModel import AttModel

dev = torch.device("cuda:0")
hp = R.make_hp()
torch.manual_seed(0)
m = AttModel(hp).to(dev).eval()
dec = m.decoder
B, T, L = 32, 500, 100
hb = torch.randn(B, T, 640, device=dev)
lengths = torch.linspace(500, 300, B).to(torch.int32).to(dev)
for b in range(B):
    hb[b, int(lengths[b]):] = 0
tgt = torch.randint(0, 1000, (B, L), device=dev)

def run():
    with torch.no_grad():
        return dec(hb, lengths, tgt)

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

eager = timeit(run)
print(f"decoder forward, kernel by kernel: {eager:.3f} ms", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        run()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = run()
graph = timeit(g.replay)
print(f"decoder forward, one graph replay:  {graph:.3f} ms", flush=True)
ref = run()
g.replay()
torch.cuda.synchronize()
print("same output:", bool(torch.equal(ref, y)))
