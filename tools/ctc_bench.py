"""CTC head at the config-2 shape: kernel times of asr_ctc_loss_fwd_bwd (HIP events on the launch
stream) against their rooflines, and the whole CTC training step (CTCModel fwd + loss + bwd)."""
import os, sys, time, types, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import time_kernel_events
from asr_amd.Loss.ctc import ctc_loss
from asr_amd.Models.CTCModel import CTCModel
from asr_amd import streams

dev = torch.device("cuda:0")
B, T, V, L = 32, 500, 1000, 100
g = torch.Generator().manual_seed(1)
z = (torch.randn(B, T, V, generator=g)).to(dev)
tgt = torch.randint(1, V, (B, L), generator=g).to(dev)
il = torch.linspace(500, 300, B).long()
tl = torch.linspace(100, 50, B).long().to(dev)
il_d = il.to(dev)
res = {}
dt = time_kernel_events(lambda: ctc_loss(z, tgt, il_d, tl), 20)
frames = int(il.sum())
res["ctc_loss_fwd_bwd_us"] = dt * 1e6
# algorithmic bytes: logits read twice (lse pass, gradient pass) + gradient written once, live frames only
res["ctc_loss_GBps_algorithmic"] = (3.0 * frames * V * 4) / dt / 1e9
hp = types.SimpleNamespace(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_hidden_nodes_encoder=320,
                           num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.2, encoder_type=None,
                           decoder_type="CTC", num_classes=V)
torch.manual_seed(1234)
m = CTCModel(hp).to(dev).train()
x = torch.randn(B, T, 240, generator=g)
for b in range(B):
    x[b, il[b]:] = 0
x = x.to(dev)
if os.environ.get("ASR_SIDE_STREAM", "1") != "0":
    streams.enable(True)

def step():
    m.zero_grad(set_to_none=True)
    y = m(x, il, tgt)
    loss = ctc_loss(y, tgt, il, tl)
    loss.backward()
    return loss

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
res["ctc_step_ms"] = ms
res["mel_frames_per_s"] = 3 * frames / (ms * 1e-3)
res["loss"] = float(loss)
print(json.dumps(res))
