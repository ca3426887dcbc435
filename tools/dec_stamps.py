"""Per-phase time stamps of the persistent decoder loop kernel at the config-2 shape (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from asr_amd.Models.decoder import Decoder
from bench import make_hp
lib = _lib.load()
for kv in os.environ.get("DEC_PACE", "").split():          # DEC_PACE="5=1 3=36": asr_attdec_pace settings for this run
    k, v = kv.split("=")
    lib.asr_attdec_pace(int(k), int(v))
dev = torch.device("cuda:0")
hp = make_hp()
torch.manual_seed(0)
dec = Decoder(hp).to(dev).train()
B, T, L = 32, 500, 100
NW = B * ((T + 63) // 64)
lengths = torch.linspace(500, 300, B).long()
h = torch.randn(B, T, 640, device=dev)
for b in range(B):
    h[b, lengths[b]:] = 0
tgt = torch.randint(0, 1000, (B, L), device=dev)
st = torch.zeros(NW, L, 10, dtype=torch.int64, device=dev)
with torch.no_grad():
    dec(h, lengths, tgt)
    lib.asr_attdec_debug_stamps(st.data_ptr())
    dec(h, lengths, tgt)
    torch.cuda.synchronize()
    lib.asr_attdec_debug_stamps(0)
s = st.cpu().double() * 0.01      # us (100 MHz)
steps = slice(5, L - 1)
tot = (s[:, steps, 9] - s[:, steps, 0])
print(f"step time (all WGs, steps 5..): mean {tot.mean():.2f} us")
def seg(i, j, sel=None):
    d = s[:, steps, j] - s[:, steps, i]
    ok = (s[:, steps, j] > 0) & (s[:, steps, i] > 0)
    if sel is not None:
        ok = ok & sel[:, None]
    return (float(d[ok].mean()), float(d[ok].max())) if ok.any() else (float("nan"), float("nan"))
wg = torch.arange(NW)
nt = (T + 63) // 64
act = ((wg % nt) * 64 < lengths[wg // nt])          # workgroups whose frame tile holds frames
cellwg = wg < 160
spwg = (wg >= 160) & (wg < 240)
for nm, i, j, sel in [("A wait sproj + window (active tiles)", 0, 1, act), ("B conv + energies + publish e (active)", 1, 2, act),
                      ("C1 wait e", 2, 3, None), ("C2 softmax + context + publish g", 3, 4, None),
                      ("D1 weights + wait g (cell WGs)", 4, 5, cellwg), ("D2 cell + publish S (cell WGs)", 5, 6, cellwg),
                      ("E1 weights + wait S (cell WGs)", 6, 7, cellwg), ("E2 W_ss S product (cell WGs)", 7, 8, cellwg),
                      ("E1 weights + wait S (sproj WGs)", 4, 7, spwg), ("E2 sproj + publish (sproj WGs)", 7, 8, spwg)]:
    m_, mx = seg(i, j, sel)
    print(f"  {nm:44s} mean {m_:6.2f} us   max {mx:6.2f} us")
# absolute timeline of one step for batch tile 0 (utterances 0..15): stamps relative to the first workgroup's stamp 0
att = (wg // nt) < 16
cell0 = cellwg & (wg % 2 == 0)
sp0 = spwg & ((wg - 160) % 2 == 0)
print("timeline of batch tile 0, us after the tile's first stamp 0 (min / mean / max over workgroups; mean over steps 5..)")
ref = s[att][:, steps, 0].min(dim=0).values            # per step
for nm, i, sel in [("0 step start (attention WGs)", 0, att), ("1 sproj in LDS (active tiles)", 1, att & act),
                   ("2 e published (active tiles)", 2, att & act), ("3 e gathered", 3, att), ("4 g published", 4, att),
                   ("5 g gathered (cell jobs)", 5, cell0), ("6 S published (cell jobs)", 6, cell0),
                   ("7 S gathered (sproj jobs)", 7, sp0), ("8 sproj published (sproj jobs)", 8, sp0),
                   ("9 step end (attention WGs)", 9, att)]:
    d = s[sel][:, steps, i] - ref[None, :]
    print(f"  {nm:36s} {float(d.min(dim=0).values.mean()):6.2f} / {float(d.mean()):6.2f} / {float(d.max(dim=0).values.mean()):6.2f}")
