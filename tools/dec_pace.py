"""Sweep of the pauses in front of the decoder loop kernels' waits (asr_attdec_pace, units of 256 cycles) at the config-2
shape (DEC_SHAPE=config5 / config4: those shapes): decoder forward and forward + backward per setting, HIP events.
usage: python tools/dec_pace.py which=v1,v2,... [which=...]     e.g.  2=0,8,12,16  3=0,20,40
Each listed site is swept with the others at their current values; the best value is kept before the next site."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from asr_amd.Models.decoder import Decoder
from bench import make_hp, time_kernel_events
lib = _lib.load()
dev = torch.device("cuda:0")
hp = make_hp()
shape = os.environ.get("DEC_SHAPE", "config2")
if shape == "config5":
    hp.num_hidden_nodes_encoder, hp.encoder_type = 256, "Conformer"
if shape == "config4":
    hp.num_classes = 7000
torch.manual_seed(0)
dec = Decoder(hp).to(dev).train()
B, T, L = {"config2": (32, 500, 100), "config5": (32, 374, 100), "config4": (32, 800, 150)}[shape]
lengths = torch.linspace(T, T * 3 // 5, B).long()
h = torch.randn(B, T, 256 if shape == "config5" else 640, device=dev)
for b in range(B):
    h[b, lengths[b]:] = 0
V = hp.num_classes
tgt = torch.randint(0, V, (B, L), device=dev)
w = torch.randn(B, L, V, device=dev)


def fwd_bwd():
    dec.zero_grad(set_to_none=True)
    hd = h.clone().requires_grad_(True)
    (dec(hd, lengths, tgt) * w).sum().backward()


def measure():
    with torch.no_grad():
        f = time_kernel_events(lambda: dec(h, lengths, tgt), 5)
    return f * 1e3, time_kernel_events(fwd_bwd, 5) * 1e3


f0, t0 = measure()
print(f"{shape}: current settings {[lib.asr_attdec_pace(i, -1) for i in range(16)]}: forward {f0:.3f} ms, forward + backward {t0:.3f} ms", flush=True)
for arg in sys.argv[1:]:
    which, vals = arg.split("=")
    which = int(which)
    best = None
    for v in [int(x) for x in vals.split(",")]:
        lib.asr_attdec_pace(which, v)
        f, t = measure()
        key = f if which < 8 else t
        print(f"  site {which} = {v:4d}: forward {f:.3f} ms, forward + backward {t:.3f} ms", flush=True)
        if best is None or key < best[0]:
            best = (key, v)
    lib.asr_attdec_pace(which, best[1])
    print(f"site {which}: keeping {best[1]}", flush=True)
f1, t1 = measure()
print(f"final settings {[lib.asr_attdec_pace(i, -1) for i in range(16)]}: forward {f1:.3f} ms, forward + backward {t1:.3f} ms")
