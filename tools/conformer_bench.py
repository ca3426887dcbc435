"""BASELINE config 5: AttModel with encoder_type='Conformer' (TransformerEncoder(80, 256, 16, 4, 0.1),
Models/AttModel.py:17), B=32, T_mel linspace(1500, 900) -> 374 subsampled frames, V=1000, L=100, decoder 320.
Times forward + label-smoothed CE + backward (dropout 0.1 as the reference trains) and the encoder alone."""
import os, sys, time, types, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd.Models.AttModel import AttModel
from asr_amd.Loss.label_smoothing import label_smoothing_loss
from asr_amd import streams

dev = torch.device("cuda:0")
B, Tm, V, L = 32, 1500, 1000, 100
layers = int(os.environ.get("CONF_LAYERS", "16"))
hp = types.SimpleNamespace(lmfb_dim=80, frame_stacking=1, num_hidden_nodes=320, num_hidden_nodes_encoder=256,
                           num_hidden_nodes_decoder=320, num_encoder_layer=5, encoder_dropout=0.2,
                           encoder_type="Conformer", decoder_type="Attention", num_classes=V, eos_id=1,
                           T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=200, score_func="log_softmax")
torch.manual_seed(1234)
m = AttModel(hp)
if layers != 16:
    m.encoder.encoder.layers = torch.nn.Sequential(*list(m.encoder.encoder.layers)[:layers])
m = m.to(dev).train()
g = torch.Generator().manual_seed(1234)
lens = torch.linspace(1500, 900, B).long()
x = torch.randn(B, Tm, 80, generator=g)
src_pos = torch.zeros(B, Tm, dtype=torch.int64)
for b in range(B):
    x[b, lens[b]:] = 0
    src_pos[b, :lens[b]] = torch.arange(1, int(lens[b]) + 1)
x, src_pos = x.to(dev), src_pos.to(dev)
tl = torch.linspace(100, 50, B).long()
tgt = torch.randint(0, V, (B, L), generator=g)
for b in range(B):
    tgt[b, tl[b]:] = 0
tgt, tl_d = tgt.to(dev), tl.to(dev)
if os.environ.get("ASR_SIDE_STREAM", "1") != "0":      # weight / bias gradients beside the data-gradient chain
    streams.enable(True)


def step():
    m.zero_grad(set_to_none=True)
    y = m(x, lens, tgt, src_pos)
    loss = label_smoothing_loss(y, tgt, tl_d, hp.T_norm, hp.B_norm)
    loss.backward()
    return loss


def enc_step():
    m.zero_grad(set_to_none=True)
    h = m.encoder(x, src_pos)
    h.sum().backward()


def timed(fn, warm, n):
    """Median of per-step HIP-event times, the host at most two steps ahead of the device (bench.py::timed_steps)."""
    from bench import timed_steps, median
    per, _, out = timed_steps(fn, n, warm)
    return median(per), out


n = int(os.environ.get("CONF_STEPS", "10"))
ms, loss = timed(step, 3, n)
res = {"workload": f"config 5: Conformer x{layers} d256 h4, B=32, T_mel linspace(1500,900) -> T'=374, V=1000, L=100, fp32, dropout 0.1",
       "step_ms": ms, "mel_frames_per_s": float(lens.sum()) / (ms * 1e-3), "loss": float(loss),
       "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
if os.environ.get("CONF_ENC_ONLY", "1") != "0":
    res["encoder_fwd_bwd_ms"], _ = timed(enc_step, 2, n)
print(json.dumps(res))
