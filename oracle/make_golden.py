"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (this container only).

    python oracle/make_golden.py            # writes tests/golden/<case>.npz

Imports ``Models.AttModel`` and ``Loss.label_smoothing`` unchanged from
/root/reference (read-only), seeds torch, and records inputs, every parameter,
``youtput``, the loss, every parameter gradient, d loss / d x and the greedy and
beam-search token sequences.  Only the resulting *data* is committed; no reference code
travels.  The reference hard-codes ``.cuda()`` in Loss/label_smoothing.py:18;
with no GPU in this container ``Tensor.cuda`` is made an identity for the
duration of the call (the reference file is untouched).

Each case is stored twice: as the reference computes it (float32) and with the
same weights/inputs promoted to float64 (a tighter anchor for error budgets).
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

CASES = {
    # name: (hp overrides, B, T_mel, mel lengths, L, text lengths, seed)
    "tiny_a": (dict(lmfb_dim=4, frame_stacking=3, num_hidden_nodes=8, num_encoder_layer=2,
                    num_classes=11), 3, 52, [52, 40, 23], 5, [5, 3, 4], 11),
    "tiny_b": (dict(lmfb_dim=6, frame_stacking=1, num_hidden_nodes=16, num_encoder_layer=3,
                    num_classes=23, B_norm=True), 4, 21, [9, 21, 14, 3], 7, [7, 2, 5, 1], 12),
    "sc_like": (dict(lmfb_dim=40, frame_stacking=3, num_hidden_nodes=32, num_encoder_layer=2,
                     num_classes=37), 2, 98, [98, 98], 3, [3, 3], 13),
    "long_t": (dict(lmfb_dim=5, frame_stacking=1, num_hidden_nodes=8, num_encoder_layer=1,
                    num_classes=7, T_norm=False), 2, 130, [130, 77], 4, [4, 2], 14),
    "mixed_h": (dict(lmfb_dim=4, frame_stacking=2, num_hidden_nodes=8, num_hidden_nodes_encoder=12,
                     num_hidden_nodes_decoder=20, num_encoder_layer=2, num_classes=13),
                5, 40, [31, 40, 8, 40, 17], 6, [6, 6, 1, 4, 3], 15),
}


def make_hp(**kw):
    d = dict(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5,
             encoder_dropout=0.0, encoder_type=None, decoder_type='Attention', num_classes=1000,
             eos_id=1, T_norm=True, B_norm=False, beam_width=1, max_decoder_seq_len=12,
             score_func='log_softmax')
    d.update(kw)
    d.setdefault('num_hidden_nodes_encoder', d['num_hidden_nodes'])
    d.setdefault('num_hidden_nodes_decoder', d['num_hidden_nodes'])
    return types.SimpleNamespace(**d)


def run_case(name, spec, AttModel, label_smoothing_loss, frame_stacking):
    over, B, T_mel, mel_len, L, text_len, seed = spec
    hp = make_hp(**over)
    torch.manual_seed(seed)
    model = AttModel(hp)
    V = hp.num_classes
    mel = torch.randn(B, T_mel, hp.lmfb_dim)
    mel_lengths = torch.tensor(mel_len, dtype=torch.int64)
    for b in range(B):
        mel[b, mel_len[b]:] = 0.0
    targets = torch.randint(0, V, (B, L), dtype=torch.int64)
    text_lengths = torch.tensor(text_len, dtype=torch.int64)
    for b in range(B):
        targets[b, text_len[b]:] = 0

    out = {"meta_hp": np.array(repr(sorted(vars(hp).items()))),
           "mel": mel.numpy(), "mel_lengths": mel_lengths.numpy(),
           "targets": targets.numpy(), "text_lengths": text_lengths.numpy()}
    for k, v in model.state_dict().items():
        out["param." + k] = v.detach().numpy().copy()

    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        # the reference allocates its decoder state with torch.zeros (default
        # dtype), so the f64 pass switches the default dtype for its duration
        torch.set_default_dtype(dtype)
        m = AttModel(hp).to(dtype)
        m.load_state_dict({k: v.to(dtype) for k, v in model.state_dict().items()})
        m.train()
        xin = mel.to(dtype)
        x, lens = frame_stacking(xin, mel_lengths, hp.frame_stacking)
        x = x.clone().requires_grad_(True)
        y = m(x, lens, targets)
        loss = label_smoothing_loss(y, targets.contiguous(), text_lengths, hp.T_norm, hp.B_norm)
        loss.backward()
        if tag == "f32":
            out["x"] = x.detach().numpy().copy()
            out["lengths"] = lens.numpy().copy()
        out[f"{tag}.youtput"] = y.detach().numpy().copy()
        out[f"{tag}.loss"] = np.array(float(loss))
        out[f"{tag}.dx"] = x.grad.numpy().copy()
        for k, p in m.named_parameters():
            out[f"{tag}.grad.{k}"] = p.grad.numpy().copy()
        torch.set_default_dtype(torch.float32)

    # greedy decode (Models/decoder.py:63-130, beam_width=1), utterance 0 and 1,
    # eos swept over the vocabulary so the raw argmax sequence is exposed.
    model.eval()
    x, lens = frame_stacking(mel, mel_lengths, hp.frame_stacking)
    for utt in range(min(B, 2)):
        res = -np.ones((V, hp.max_decoder_seq_len), dtype=np.int64)
        for eos in range(V):
            hp.eos_id = eos
            r = model.decode(x[utt:utt + 1], lens[utt:utt + 1], None, None)
            res[eos, :len(r)] = r
        out[f"greedy.utt{utt}"] = res
    out["greedy.max_len"] = np.array(hp.max_decoder_seq_len)
    # beam search (the same Models/decoder.py:63-130 with beam_width > 1): width 3 with the
    # log_softmax score on utterances 0 and 1, width 2 with the softmax score on utterance 0;
    # eos swept as above (decode() returns the best hypothesis once it ends in eos, else []).
    for tag, width, score, utts in (("beam3", 3, "log_softmax", range(min(B, 2))), ("beam2sm", 2, "softmax", (0,))):
        hp.beam_width, hp.score_func = width, score
        for utt in utts:
            res = -np.ones((V, hp.max_decoder_seq_len), dtype=np.int64)
            for eos in range(V):
                hp.eos_id = eos
                r = model.decode(x[utt:utt + 1], lens[utt:utt + 1], None, None)
                res[eos, :len(r)] = r
            out[f"{tag}.utt{utt}"] = res
    hp.beam_width, hp.score_func = 1, "log_softmax"
    return out


def main():
    sys.path.insert(0, REF)
    from Models.AttModel import AttModel              # noqa: E402  (the reference)
    from Loss.label_smoothing import label_smoothing_loss  # noqa: E402
    from utils.utils import frame_stacking            # noqa: E402
    torch.Tensor.cuda = lambda self, *a, **k: self     # label_smoothing.py:18 on a CPU-only box
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)
    for name, spec in CASES.items():
        data = run_case(name, spec, AttModel, label_smoothing_loss, frame_stacking)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: loss f32 {float(data['f32.loss']):.6f} f64 {float(data['f64.loss']):.6f} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
