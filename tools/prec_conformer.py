"""Config-5 step (dropout off) under the Conformer encoder's arithmetic settings: loss and every gradient with the
16 layers' products in the bf16x3 split (Conv2d subsampling exact, or split too) against the all-exact-fp32 step.
GPU box only."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import _lib
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd.Models import Transformer_encoder as TE
from asr_amd.Models.AttModel import AttModel

lib = _lib.load()
dev = torch.device("cuda:0")
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
args = (x.to(dev), lens, tgt.to(dev), src_pos.to(dev))
tgt, tl = tgt.to(dev), tl.to(dev)


def run(layers, cnn, rest):
    TE.GEMM_PRECISION, TE.CNN_FWD_GEMM_PRECISION = layers, cnn
    lib.asr_gemm_precision({"f32": 0, "bf16x3": 1}[rest])
    m.zero_grad(set_to_none=True)
    y = m(*args)
    loss = label_smoothing_loss(y, tgt, tl, True, False)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters()}


l0, g0 = run("f32", "f32", "f32")
gmax = max(float(v.abs().max()) for v in g0.values())
# biases in front of a BatchNorm have an exactly-zero gradient in exact arithmetic: round-off only, left out
noise = {k for k, v in g0.items() if float(v.abs().max()) < 1e-5 * gmax}
print(f"largest gradient entry {gmax:.3e}; {len(noise)} tensors are round-off only (< 1e-5 of it):", sorted(noise)[:3], "...")
for layers, cnn, rest in (("f32", "f32", "bf16x3"), ("bf16x3", "f32", "bf16x3"), ("bf16x3", "bf16x3", "bf16x3")):
    l, gr = run(layers, cnn, rest)
    errs = {k: float((gr[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-30) for k in g0 if k not in noise}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    n_over = sum(v > 1e-3 for v in errs.values())
    print(f"layers and all backward products {layers}, Conv2d subsampling forward {cnn}, decoder {rest}: loss rel {abs(l - l0) / abs(l0):.2e}; tensors over 1e-3: {n_over} of "
          f"{len(errs)}; worst " + ", ".join(f"{k} {v:.1e}" for k, v in worst), flush=True)
