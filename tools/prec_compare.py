"""Full-size steps (config 2, config 5; dropout off) in the three GEMM arithmetic modes: loss and every gradient
of mode 1 (bf16x3) and mode 2 (bf16) against mode 0 (exact fp32 products, the mode the oracle pins).  GPU box only."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd.Models.AttModel import AttModel
from bench import make_batch, make_hp

lib = _lib.load()
dev = torch.device("cuda:0")


def run(m, args, tgt, tl, mode):
    lib.asr_gemm_precision(mode)
    m.zero_grad(set_to_none=True)
    y = m(*args)
    loss = label_smoothing_loss(y, tgt, tl, True, False)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}


def report(name, m, args, tgt, tl):
    l0, y0, g0 = run(m, args, tgt, tl, 0)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for mode, tag in ((1, "bf16x3"), (2, "bf16")):
        l, y, g = run(m, args, tgt, tl, mode)
        errs = {k: float((g[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-30) for k in g0}
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        floor = {k: float((g[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 0.05 * gmax) for k in g0}
        print(f"{name} {tag}: loss {l:.6f} vs {l0:.6f} (rel {abs(l - l0) / abs(l0):.2e})  youtput rel "
              f"{float((y - y0).abs().max() / y0.abs().max()):.2e}  worst grads (own max) "
              + ", ".join(f"{k} {v:.1e}" for k, v in worst) + f"  worst with 5% floor {max(floor.values()):.1e}", flush=True)
    lib.asr_gemm_precision(1)


hp = make_hp(encoder_dropout=0.0)
torch.manual_seed(1234)
m = AttModel(hp).to(dev).train()
x, lengths, tgt, tl = make_batch(hp, dev)
report("config 2", m, (x, lengths, tgt), tgt, tl)
del m
B, Tm, V, L = 32, 1500, 1000, 100
hp5 = types.SimpleNamespace(lmfb_dim=80, frame_stacking=1, num_hidden_nodes=320, num_hidden_nodes_encoder=256,
                            num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.0,
                            encoder_type="Conformer", decoder_type="Attention", num_classes=V, eos_id=1,
                            T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=200, score_func="log_softmax")
torch.manual_seed(1234)
m = AttModel(hp5)
for mod in m.modules():
    if hasattr(mod, "p") and isinstance(getattr(mod, "p"), float):
        mod.p = 0.0
m = m.to(dev).train()
g = torch.Generator().manual_seed(1234)
lens = torch.linspace(1500, 900, B).long()
x = torch.randn(B, Tm, 80, generator=g)
src_pos = torch.zeros(B, Tm, dtype=torch.int64)
for b in range(B):
    x[b, lens[b]:] = 0
    src_pos[b, :lens[b]] = torch.arange(1, int(lens[b]) + 1)
tl = torch.linspace(100, 50, B).long()
tgt = torch.randint(0, V, (B, L), generator=g)
for b in range(B):
    tgt[b, tl[b]:] = 0
report("config 5", m, (x.to(dev), lens, tgt.to(dev), src_pos.to(dev)), tgt.to(dev), tl.to(dev))
