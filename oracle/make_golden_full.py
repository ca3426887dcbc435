"""Generate tests/golden/full/*.npz: DIGESTS of full-size (BASELINE config 2) training steps of THE REFERENCE.

    python oracle/make_golden_full.py [init peaked]     # this container only; ~10 minutes on 8 threads

Imports ``Models.AttModel``, ``Loss.label_smoothing`` and ``utils.utils.frame_stacking`` unchanged from
/root/reference (read-only), builds the model under torch.manual_seed(1234) with the reference's own default
initialisation, runs forward + label-smoothed CE + backward on the synthetic config-2 batch (B=32, T_mel 1500..900,
L=100, V=1000, dropout 0) and stores the digest defined in tests/_full_digest.py: loss, 4096 sampled youtput
entries, per-gradient L2 norm / largest entry / 8 projections, the same for d loss / d x, and a checksum of every
parameter.  Two cases: "init" (the default initialisation, what bench.py times) and "peaked" (the same weights
rescaled so that the alignment is peaked and the attention's gradients are not degenerate; the rule is
tests/_full_digest.py::peaked_scale).  Each case as the reference computes it (float32) and promoted to float64.
Only data is committed; no reference code travels.  ``Tensor.cuda`` is an identity for the call
(Loss/label_smoothing.py:18 hard-codes it; no GPU here); the reference's files are untouched.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from tests import _full_digest as D      # noqa: E402

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "..", "tests", "golden", "full")


def main():
    cases = sys.argv[1:] or ["init", "peaked"]
    sys.path.insert(0, REF)
    from Models.AttModel import AttModel                     # noqa: E402  (the reference)
    from Loss.label_smoothing import label_smoothing_loss    # noqa: E402
    from utils.utils import frame_stacking                   # noqa: E402
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    hp = D.make_hp()
    torch.manual_seed(D.SEED)
    base = AttModel(hp)
    base_state = {k: v.detach().clone() for k, v in base.state_dict().items()}
    mel, mel_len, tgt, text_len = D.make_inputs(hp)
    for case in cases:
        state = D.scaled_state(base_state, case)
        out = {}
        for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
            t0 = time.time()
            torch.set_default_dtype(dtype)       # the reference allocates its decoder state with torch.zeros
            m = AttModel(hp).to(dtype)
            m.load_state_dict({k: v.to(dtype) for k, v in state.items()})
            m.train()
            x, lens = frame_stacking(mel.to(dtype), mel_len, hp.frame_stacking)
            x = x.clone().requires_grad_(True)
            y = m(x, lens, tgt)
            loss = label_smoothing_loss(y, tgt.contiguous(), text_len, hp.T_norm, hp.B_norm)
            loss.backward()
            torch.set_default_dtype(torch.float32)
            names = [k for k, _ in m.named_parameters()]
            ps = [p for _, p in m.named_parameters()]
            dg = D.step_digest(names, ps, [p.grad for p in ps], y, loss.detach(), x.grad)
            for k, v in dg.items():
                out[f"{tag}.{k}"] = v
            print(f"{case} {tag}: loss {float(loss):.6f}  ({time.time() - t0:.0f} s)", flush=True)
        path = os.path.join(OUT, f"config2_{case}.npz")
        np.savez_compressed(path, **out)
        print(f"-> {path} {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


if __name__ == "__main__":
    main()
