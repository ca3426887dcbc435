"""Decoder forward and forward + backward at the config-2 shape (DEC_SHAPE=config5: the config-5 shape): the persistent loop kernels against the
per-step launches (HIP events; the backward figure includes the hoisted GEMMs and phase 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from asr_amd.Models.decoder import Decoder
from bench import make_hp, time_kernel_events
lib = _lib.load()
dev = torch.device("cuda:0")
hp = make_hp()
# DEC_SHAPE=config5: the Conformer branch's decoder input (T' = 374 subsampled frames of width 256)
c5 = os.environ.get("DEC_SHAPE") == "config5"
if c5:
    hp.num_hidden_nodes_encoder, hp.encoder_type = 256, "Conformer"
torch.manual_seed(0)
dec = Decoder(hp).to(dev).train()
B, T, L = (32, 374, 100) if c5 else (32, 500, 100)
lengths = torch.linspace(T, T * 3 // 5, B).long()
h = torch.randn(B, T, 256 if c5 else 640, device=dev)
for b in range(B):
    h[b, lengths[b]:] = 0
tgt = torch.randint(0, 1000, (B, L), device=dev)
w = torch.randn(B, L, 1000, device=dev)


def fwd_bwd():
    dec.zero_grad(set_to_none=True)
    hd = h.clone().requires_grad_(True)
    (dec(hd, lengths, tgt) * w).sum().backward()


for on in (0, 1, 0, 1):
    lib.asr_attdec_persist(on)
    with torch.no_grad():
        dt = time_kernel_events(lambda: dec(h, lengths, tgt), 5)
    dt2 = time_kernel_events(fwd_bwd, 5)
    print(f"persistent={on}: decoder forward {dt*1e3:.3f} ms ({dt*1e6/L:.1f} us per label step incl. hoisted GEMMs); "
          f"forward + backward {dt2*1e3:.3f} ms -> backward {(dt2-dt)*1e3:.3f} ms ({(dt2-dt)*1e6/L:.1f} us per label step "
          f"incl. hoisted GEMMs, phase 2, weight gradients)", flush=True)
