"""Per-phase time stamps of the persistent decoder BACKWARD loop kernel at the config-2 shape (diagnostics)."""
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
nt = (T + 63) // 64
NW = B * nt
lengths = torch.linspace(500, 300, B).long()
h = torch.randn(B, T, 640, device=dev)
for b in range(B):
    h[b, lengths[b]:] = 0
tgt = torch.randint(0, 1000, (B, L), device=dev)
wgt = torch.randn(B, L, 1000, device=dev)
st = torch.zeros(NW, L, 10, dtype=torch.int64, device=dev)
lib.asr_attdec_persist(1)
for it in range(2):
    dec.zero_grad(set_to_none=True)
    hd = h.clone().requires_grad_(True)
    y = dec(hd, lengths, tgt)
    torch.cuda.synchronize()
    if it == 1:
        lib.asr_attdec_debug_stamps(st.data_ptr())
    (y * wgt).sum().backward()
    torch.cuda.synchronize()
    lib.asr_attdec_debug_stamps(0)
s = st.cpu().double() * 0.01      # us (100 MHz); index [wg][label step s][stamp]; the kernel walks s = L-1 .. 0
s = s.flip(1)                     # -> order of execution
steps = slice(5, L - 2)
wg = torch.arange(NW)
act = ((wg % nt) * 64 < lengths[wg // nt])
# job roles are dealt by load (attdec_bwd_persist_kernel): rank by (frames in the tile, index)
nrows_wg = (lengths[wg // nt] - (wg % nt) * 64).clamp(0, 64)
order = torch.argsort(nrows_wg * 4096 + wg)
rank = torch.empty_like(order)
rank[order] = torch.arange(NW)
xwg = rank < 40
gwg = (rank >= 40) & (rank < 200)
tot = s[:, steps, 9] - s[:, steps, 0]
print(f"step time (all WGs): mean {tot.mean():.2f} us")
def seg(i, j, sel):
    d = s[sel][:, steps, j] - s[sel][:, steps, i]
    ok = (s[sel][:, steps, j] > 0) & (s[sel][:, steps, i] > 0)
    return float(d[ok].mean()), float(d[ok].max())
for nm, i, j, sel in [("X  wait dsp + gather (X WGs)", 0, 1, xwg), ("X  product + cell bwd + publish drec", 1, 2, xwg),
                      ("G  wait drec + gather (G WGs)", 0, 3, gwg), ("G  product + publish dg partial", 3, 4, gwg),
                      ("DA wait dg (active)", 5, 6, act), ("DA dalpha + dot exchange + de", 6, 7, act),
                      ("EB energy backward + publish", 7, 8, act), ("RD wait partials + reduce + publish dsp", 8, 9, act)]:
    m_, mx = seg(i, j, sel)
    print(f"  {nm:44s} mean {m_:6.2f} us   max {mx:6.2f} us")
# timeline of batch tile 0 relative to the first X job's stamp 0
x0 = xwg & (rank % 2 == 0)
g0 = gwg & (((rank - 40) >> 1) % 2 == 0)
att = ((wg // nt) < 16) & act
ref = s[x0][:, steps, 0].min(dim=0).values
print("timeline of batch tile 0, us after its first X job's stamp 0 (min / mean / max over workgroups)")
for nm, i, sel in [("0 step start (X jobs)", 0, x0), ("1 dsp gathered (X jobs)", 1, x0), ("2 drec published (X jobs)", 2, x0),
                   ("3 drec gathered (G jobs)", 3, g0), ("4 dg partial published (G jobs)", 4, g0),
                   ("5 conv^T done (active tiles)", 5, att), ("6 dg gathered", 6, att), ("7 de done", 7, att),
                   ("8 dsp partial / dF published", 8, att), ("9 dsp chunk published = step end", 9, att)]:
    d = s[sel][:, steps, i] - ref[None, :]
    print(f"  {nm:40s} {float(d.min(dim=0).values.mean()):6.2f} / {float(d.mean()):6.2f} / {float(d.max(dim=0).values.mean()):6.2f}")
