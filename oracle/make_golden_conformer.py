"""Generate tests/golden/conformer/*.npz by running THE REFERENCE's Conformer encoder (this container only).

    python oracle/make_golden_conformer.py

Imports ``Models.Transformer_encoder.TransformerEncoder`` unchanged from /root/reference and
constructs it through its own arguments ``(src_vocab, d_model, N_e, heads, dropout)`` at small
sizes (AttModel.py:17 hard-codes (lmfb_dim, 256, 16, 4, 0.1): 48 M parameters, too large for a
fixture).  dropout = 0 so the training-mode pass is deterministic; BatchNorm then uses batch
statistics, the eval-mode pass its running statistics.  Recorded per case, in float32 as the
reference computes it and with everything promoted to float64: inputs (src, src_pos), every
parameter and buffer, the eval-mode output, the training-mode output, the gradient of
``sum(output * R)`` (R stored) w.r.t. every parameter, and the BatchNorm running statistics after
that one training-mode forward.  Only data is committed.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "conformer")

CASES = {
    # name: (idim, d_model, N_e, heads, B, T, lengths, seed)
    "conf_tiny": (20, 16, 1, 2, 3, 45, [45, 33, 21], 21),
    "conf_small": (24, 32, 2, 4, 2, 70, [70, 52], 22),
    "conf_full_len": (17, 24, 1, 3, 2, 41, [41, 41], 23),
}


def run_case(spec, TransformerEncoder):
    idim, d, N, heads, B, T, lens, seed = spec
    torch.manual_seed(seed)
    model = TransformerEncoder(idim, d, N, heads, 0.0)
    # BatchNorm/LayerNorm start at weight 1 / bias 0: perturb them so their gradients' paths are exercised
    with torch.no_grad():
        for k, p in model.named_parameters():
            if "norm" in k:
                p.add_(0.1 * torch.randn_like(p))
    src = torch.randn(B, T, idim)
    src_pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        src[b, n:] = 0.0
        src_pos[b, :n] = torch.arange(1, n + 1)
    out = {"meta": np.array(repr(dict(idim=idim, d_model=d, N_e=N, heads=heads))),
           "src": src.numpy(), "src_pos": src_pos.numpy()}
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in state0.items():
        out["param." + k] = v.numpy().copy()
    R = None
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        torch.set_default_dtype(dtype)
        m = TransformerEncoder(idim, d, N, heads, 0.0).to(dtype)
        m.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in state0.items()})
        m.encoder.pe.pe = m.encoder.pe.pe.to(dtype) if dtype == torch.float64 else m.encoder.pe.pe
        m.eval()
        with torch.no_grad():
            out[f"{tag}.eval_out"] = m(src.to(dtype), src_pos).numpy().copy()
        m.train()
        y = m(src.to(dtype), src_pos)
        if R is None:
            R = torch.randn(*y.shape)
            out["R"] = R.numpy()
        (y * R.to(dtype)).sum().backward()
        out[f"{tag}.train_out"] = y.detach().numpy().copy()
        for k, p in m.named_parameters():
            # d_ff = 2048 is hard-coded (Transformer_encoder.py:192): the float64 copies of the large
            # gradients would triple the fixture, so float64 keeps the small tensors only
            if tag == "f32" or p.numel() <= 4096:
                out[f"{tag}.grad.{k}"] = p.grad.numpy().copy()
        for k, v in m.state_dict().items():
            if "running_" in k or "num_batches" in k:
                out[f"{tag}.after.{k}"] = v.numpy().copy()
        torch.set_default_dtype(torch.float32)
    return out


def main():
    sys.path.insert(0, REF)
    from Models.Transformer_encoder import TransformerEncoder     # noqa: E402  (the reference)
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)
    for name, spec in CASES.items():
        data = run_case(spec, TransformerEncoder)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: out {data['f32.train_out'].shape} |f32-f64| "
              f"{np.abs(data['f32.train_out'] - data['f64.train_out']).max():.2e} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
