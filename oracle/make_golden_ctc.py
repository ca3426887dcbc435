"""Generate tests/golden/ctc/*.npz by running THE REFERENCE's CTC path (this container only).

    python oracle/make_golden_ctc.py

Imports ``Models.CTCModel.CTCModel`` unchanged from /root/reference (read-only), runs
``forward``, the two loss lines of train.py:56-57 and ``backward``, and records inputs, every
parameter, the logits, the loss, every gradient and the greedy ``decode`` output.  Only the
resulting data is committed.  ``CTCModel.decode`` reads a module-global ``hp`` whose import the
reference comments out (Models/CTCModel.py:6,31): the generator sets that attribute on the
imported module object for the duration of the call (the file is untouched).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ctc")

CASES = {
    # name: (hp overrides, B, T_mel, mel lengths, L, text lengths, seed, repeat-heavy labels)
    "ctc_tiny": (dict(lmfb_dim=4, frame_stacking=3, num_hidden_nodes=8, num_encoder_layer=2, num_classes=11),
                 3, 52, [52, 40, 23], 5, [5, 3, 4], 21, False),
    "ctc_repeat": (dict(lmfb_dim=6, frame_stacking=1, num_hidden_nodes=12, num_encoder_layer=1, num_classes=5),
                   4, 30, [30, 21, 14, 30], 7, [7, 2, 5, 0], 22, True),
    "ctc_sc_like": (dict(lmfb_dim=40, frame_stacking=3, num_hidden_nodes=32, num_encoder_layer=2, num_classes=37),
                    2, 98, [98, 98], 3, [3, 3], 23, False),
    "ctc_unsorted": (dict(lmfb_dim=5, frame_stacking=2, num_hidden_nodes=8, num_encoder_layer=2, num_classes=40),
                     5, 80, [31, 80, 24, 80, 57], 9, [6, 9, 1, 4, 3], 24, False),
}


def make_hp(**kw):
    d = dict(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5, encoder_dropout=0.0,
             encoder_type=None, decoder_type='CTC', num_classes=1000)
    d.update(kw)
    d.setdefault('num_hidden_nodes_encoder', d['num_hidden_nodes'])
    d.setdefault('num_hidden_nodes_decoder', d['num_hidden_nodes'])
    return types.SimpleNamespace(**d)


def run_case(spec, CTCModel, frame_stacking, ctc_module):
    over, B, T_mel, mel_len, L, text_len, seed, repeats = spec
    hp = make_hp(**over)
    torch.manual_seed(seed)
    model = CTCModel(hp)
    V = hp.num_classes
    mel = torch.randn(B, T_mel, hp.lmfb_dim)
    mel_lengths = torch.tensor(mel_len, dtype=torch.int64)
    for b in range(B):
        mel[b, mel_len[b]:] = 0.0
    hi = 3 if repeats else V                 # labels 1..2: many repeats, skip transitions forbidden
    targets = torch.randint(1, hi, (B, L), dtype=torch.int64)
    text_lengths = torch.tensor(text_len, dtype=torch.int64)
    for b in range(B):
        targets[b, text_len[b]:] = 0
    out = {"meta_hp": np.array(repr(sorted(vars(hp).items()))), "mel": mel.numpy(),
           "mel_lengths": mel_lengths.numpy(), "targets": targets.numpy(), "text_lengths": text_lengths.numpy()}
    for k, v in model.state_dict().items():
        out["param." + k] = v.detach().numpy().copy()
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        m = CTCModel(hp).to(dtype)
        m.load_state_dict({k: v.to(dtype) for k, v in model.state_dict().items()})
        m.train()
        x, lens = frame_stacking(mel.to(dtype), mel_lengths, hp.frame_stacking)
        x = x.clone().requires_grad_(True)
        y = m(x, lens, targets, None)
        y.retain_grad()
        lp = F.log_softmax(y, dim=2).transpose(0, 1)                       # train.py:56
        loss = F.ctc_loss(lp, targets, lens, text_lengths, blank=0)       # train.py:57
        loss.backward()
        if tag == "f32":
            out["x"] = x.detach().numpy().copy()
            out["lengths"] = lens.numpy().copy()
        out[f"{tag}.youtput"] = y.detach().numpy().copy()
        out[f"{tag}.loss"] = np.array(float(loss))
        out[f"{tag}.dyoutput"] = y.grad.numpy().copy()
        out[f"{tag}.dx"] = x.grad.numpy().copy()
        for k, p in m.named_parameters():
            out[f"{tag}.grad.{k}"] = p.grad.numpy().copy()
    model.eval()
    x, lens = frame_stacking(mel, mel_lengths, hp.frame_stacking)
    ctc_module.hp = hp
    try:
        res = model.decode(x, lens)
    finally:
        del ctc_module.hp
    width = max(1, max(len(r) for r in res))
    dec = -np.ones((B, width), dtype=np.int64)
    for b, r in enumerate(res):
        dec[b, :len(r)] = r
    out["decode"] = dec
    return out


def main():
    sys.path.insert(0, REF)
    import Models.CTCModel as ctc_module                 # noqa: E402  (the reference)
    from utils.utils import frame_stacking               # noqa: E402
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)
    for name, spec in CASES.items():
        data = run_case(spec, ctc_module.CTCModel, frame_stacking, ctc_module)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: loss f32 {float(data['f32.loss']):.6f} f64 {float(data['f64.loss']):.6f} "
              f"decode {data['decode'].tolist()} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
