"""Per-phase time stamps of the persistent decoder loop kernel at the config-2 shape (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from asr_amd.Models.decoder import Decoder
from bench import make_hp
lib = _lib.load()
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
names = ["A wait sproj+load", "B energy", "publish e + wait e", "C softmax+ctx", "publish g + wait g,S (cell WGs)",
         "D cell compute", "publish S + wait S (sproj WGs)", "E sproj compute", "rest of step"]
steps = slice(5, L - 1)
tot = (s[:, steps, 9] - s[:, steps, 0])
print(f"step time (all WGs, steps 5..): mean {tot.mean():.2f} us")
def seg(i, j, sel=None):
    d = s[:, steps, j] - s[:, steps, i]
    ok = (s[:, steps, j] > 0) & (s[:, steps, i] > 0)
    if sel is not None:
        ok = ok & sel[:, None]
    return float(d[ok].mean()) if ok.any() else float("nan"), float(d[ok].max()) if ok.any() else float("nan")
wg = torch.arange(NW)
cellwg = wg < 160
spwg = (wg >= 160) & (wg < 240)
for nm, i, j, sel in [(names[0], 0, 1, None), (names[1], 1, 2, None), (names[2], 2, 3, None), (names[3], 3, 4, None),
                      (names[4], 4, 5, cellwg), (names[5], 5, 6, cellwg), ("publish S .. end of step (cell WGs)", 6, 9, cellwg),
                      ("publish g .. S ready (sproj WGs)", 4, 7, spwg), (names[7], 7, 8, spwg), ("publish sproj .. end", 8, 9, spwg)]:
    m, mx = seg(i, j, sel)
    print(f"  {nm:44s} mean {m:6.2f} us   max {mx:6.2f} us")
