"""600 training steps (forward, loss, backward, fused clip + Adam) on one config-2 batch with the side stream on: the loss falls,
stays finite, no in-launch wait times out (17.5 ms per full step).  usage: python tools/soak.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_hp, make_batch
from asr_amd.Models.AttModel import AttModel
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd.optim import ClipAdam
from asr_amd.parallel import BucketedGradReducer
from asr_amd.Models import encoder as E
from asr_amd import streams
dev = torch.device("cuda:0")
streams.prepare(dev)
hp = make_hp(encoder_dropout=0.2)
torch.manual_seed(1)
m = AttModel(hp).to(dev).train()
red = BucketedGradReducer(m)          # enables the side stream as the bench does
x, lengths, tgt, tl = make_batch(hp, dev, seed=3)
opt = ClipAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
losses = []
t0 = time.time()
for i in range(600):
    for p in m.parameters():
        p.grad = None
    loss = label_smoothing_loss(m(x, lengths, tgt), tgt, tl, hp.T_norm, hp.B_norm)
    loss.backward()
    red.finish()
    opt.step(max_norm=hp.clip)
    if i % 100 == 0 or i == 599:
        losses.append(float(loss))
        print(i, losses[-1], f"{time.time()-t0:.1f}s", flush=True)
torch.cuda.synchronize()
assert all(l == l for l in losses) and losses[-1] < 0.5 * losses[0], losses
assert not E.persistent_error_flags()
print("soak ok", losses)
