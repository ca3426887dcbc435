"""One rank of tests/test_gpu_dp2.py: every rank sits on GPU 0 (gloo between them), takes its share of the
global batch, runs forward + loss + backward through the bucketed reducer and writes its averaged gradients."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

import numpy as np
import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(90, exit=True)
out_dir = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from asr_amd import streams
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd.Models.AttModel import AttModel
from asr_amd.parallel import BucketedGradReducer
from tests.test_gpu_dp2 import make_case

streams.prepare(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)
hp, state, x, lengths, targets, tl = make_case()
m = AttModel(hp)
if rank == 0:
    m.load_state_dict(state)          # the other ranks keep their own random weights until the broadcast
m = m.to(dev).train()
per = x.shape[0] // world
sl = slice(rank * per, (rank + 1) * per)
with BucketedGradReducer(m) as red:
    assert red.active and red.world == world
    red.broadcast_parameters()
    for it in range(2):               # twice: the second pass reuses the flat buffers and the pinned staging
        m.zero_grad(set_to_none=True)
        loss = label_smoothing_loss(m(x[sl].to(dev), lengths[sl], targets[sl].to(dev)), targets[sl].to(dev),
                                    tl[sl].to(dev), hp.T_norm, hp.B_norm)
        loss.backward()
        red.finish()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=float(loss.detach()),
             **{k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
dist.barrier()
dist.destroy_process_group()
