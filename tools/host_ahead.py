"""Host-side time of each phase of the step (no synchronisation inside): is the host ahead of the GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_hp, make_batch
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd.Models.AttModel import AttModel
from asr_amd import streams
dev = torch.device("cuda:0")
hp = make_hp(encoder_dropout=0.2)
torch.manual_seed(1234)
model = AttModel(hp).to(dev).train()
streams.enable(True)
x, lengths, tgt, tl = make_batch(hp, dev, seed=1234)
def step(rec=None):
    t0 = time.perf_counter()
    for p in model.parameters():
        p.grad = None
    y = model(x, lengths, tgt)
    t1 = time.perf_counter()
    loss = label_smoothing_loss(y, tgt, tl, hp.T_norm, hp.B_norm)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    if rec is not None:
        rec.append((t1 - t0, t2 - t1, t3 - t2))
for _ in range(5):
    step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(10):
    step(rec)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
f = sum(r[0] for r in rec) / 10 * 1e3; l = sum(r[1] for r in rec) / 10 * 1e3; b = sum(r[2] for r in rec) / 10 * 1e3
print(f"host per step: forward {f:.2f} ms, loss {l:.2f} ms, backward {b:.2f} ms; host total {th/10*1e3:.2f} ms, GPU-inclusive {tg/10*1e3:.2f} ms")
