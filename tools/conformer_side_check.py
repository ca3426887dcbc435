"""Conformer step (CONF_LAYERS layers, config-5 widths) with the weight / bias gradients on the side stream against the
same step on one stream: same seeds, every gradient compared (split-K products add with float atomics, so the two
runs agree to rounding, not bit for bit)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd.Models.AttModel import AttModel
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd import streams

dev = torch.device("cuda:0")
B, Tm, V, L = 8, 600, 1000, 40
layers = int(os.environ.get("CONF_LAYERS", "3"))
hp = types.SimpleNamespace(lmfb_dim=80, frame_stacking=1, num_hidden_nodes=320, num_hidden_nodes_encoder=256,
                           num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.2,
                           encoder_type="Conformer", decoder_type="Attention", num_classes=V, eos_id=1,
                           T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=200, score_func="log_softmax")
torch.manual_seed(1234)
m = AttModel(hp)
m.encoder.encoder.layers = torch.nn.Sequential(*list(m.encoder.encoder.layers)[:layers])
m = m.to(dev).train()
g = torch.Generator().manual_seed(1)
lens = torch.linspace(Tm, Tm * 3 // 5, B).long()
x = torch.randn(B, Tm, 80, generator=g)
src_pos = torch.zeros(B, Tm, dtype=torch.int64)
for b in range(B):
    x[b, lens[b]:] = 0
    src_pos[b, :lens[b]] = torch.arange(1, int(lens[b]) + 1)
tl = torch.linspace(L, L // 2, B).long()
tgt = torch.randint(0, V, (B, L), generator=g)
for b in range(B):
    tgt[b, tl[b]:] = 0
x, src_pos, tgt, tl_d = x.to(dev), src_pos.to(dev), tgt.to(dev), tl.to(dev)


def grads(side):
    streams.enable(side)
    torch.manual_seed(7)                      # the dropout seeds are drawn from torch's generator
    m.zero_grad(set_to_none=True)
    loss = label_smoothing_loss(m(x, lens, tgt, src_pos), tgt, tl_d, hp.T_norm, hp.B_norm)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


l0, g0 = grads(False)
for rep in range(3):
    l1, g1 = grads(True)
    worst = max(((g1[n] - g0[n]).abs().max().item() / (g0[n].abs().max().item() + 1e-30), n) for n in g0)
    print(f"rep {rep}: loss {l0:.6f} / {l1:.6f}; {len(g0)} gradients, worst relative difference {worst[0]:.2e} ({worst[1]})", flush=True)
    assert set(g0) == set(g1) and worst[0] < 1e-5 and all(torch.isfinite(v).all() for v in g1.values())
print("ok")
