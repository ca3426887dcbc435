import sys, types, torch
sys.path.insert(0, '.')
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd.Models.AttModel import AttModel
dev = torch.device('cuda:0')
B, Tm, V, L = 32, 1500, 1000, 100
hp = types.SimpleNamespace(lmfb_dim=80, frame_stacking=1, num_hidden_nodes=320, num_hidden_nodes_encoder=256, num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.2, encoder_type="Conformer", decoder_type="Attention", num_classes=V, eos_id=1, T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=200, score_func="log_softmax")
torch.manual_seed(1234)
m = AttModel(hp).to(dev).train()
g = torch.Generator().manual_seed(1234)
lens = torch.linspace(1500, 900, B).long()
x = torch.randn(B, Tm, 80, generator=g); src_pos = torch.zeros(B, Tm, dtype=torch.int64)
for b in range(B):
    x[b, lens[b]:] = 0; src_pos[b, :lens[b]] = torch.arange(1, int(lens[b]) + 1)
tl = torch.linspace(100, 50, B).long(); tgt = torch.randint(0, V, (B, L), generator=g)
for b in range(B): tgt[b, tl[b]:] = 0
x, src_pos, tgt, tl_d = x.to(dev), src_pos.to(dev), tgt.to(dev), tl.to(dev)
def loss_of():
    torch.manual_seed(99)
    return label_smoothing_loss(m(x, lens, tgt, src_pos), tgt, tl_d, hp.T_norm, hp.B_norm)
loss = loss_of(); loss.backward()
print('loss', float(loss), 'repeat', float(loss_of().detach()))
names = [k for k, _ in m.named_parameters()]
groups = {n: (lambda k, n=n: k == n) for n in ('encoder.cnn_encoder.conv.0.weight', 'encoder.cnn_encoder.conv.0.bias', 'encoder.cnn_encoder.conv.2.weight', 'encoder.cnn_encoder.conv.2.bias', 'encoder.cnn_encoder.out.0.weight')}
print([k for k in names if not k.startswith('decoder')][:8])
with torch.no_grad():
    torch.manual_seed(7)
    dirs = {k: torch.randn_like(p) for k, p in m.named_parameters()}
    for gname, sel in groups.items():
        ks = [k for k in names if sel(k)]
        if not ks: continue
        P = dict(m.named_parameters())
        analytic = sum(float((P[k].grad.double() * dirs[k].double()).sum()) for k in ks)
        out = []
        for eps in (4e-3, 2e-3, 1e-3, 5e-4, 2.5e-4):
            vals = []
            for sgn in (+1, -1):
                for k in ks: P[k].add_(dirs[k], alpha=sgn * eps)
                vals.append(float(loss_of()))
                for k in ks: P[k].add_(dirs[k], alpha=-sgn * eps)
            out.append((vals[0] - vals[1]) / (2 * eps))
        print(gname, len(ks), 'analytic', analytic, 'fd', out, flush=True)
