"""Shared helpers for the tests: golden-fixture loading and error metrics."""
import ast
import glob
import os
import types

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.splitext(os.path.basename(p))[0]
               for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


CTC_CASES = sorted(os.path.splitext(os.path.basename(p))[0]
                   for p in glob.glob(os.path.join(GOLDEN_DIR, "ctc", "*.npz")))


V2_RUNS = (("v2beam3", 3, "log_softmax"), ("v2beam2sm", 2, "softmax"), ("v2beam4", 4, "log_softmax"),
           ("v2beam6sm", 6, "softmax"))
V2_CASES = sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "v2", "*.npz")))


def load_v2(name):
    """decode_v2 tables of the reference (oracle/make_golden_v2.py) for the case ``name``."""
    return np.load(os.path.join(GOLDEN_DIR, "v2", name + ".npz"), allow_pickle=False)


def load_case(name):
    sub = "ctc" if name.startswith("ctc_") else ""
    z = np.load(os.path.join(GOLDEN_DIR, sub, name + ".npz"), allow_pickle=False)
    hp = types.SimpleNamespace(**dict(ast.literal_eval(str(z["meta_hp"]))))
    return z, hp


def params_of(z, dtype=torch.float32):
    return {k[len("param."):]: torch.from_numpy(z[k]).to(dtype)
            for k in z.files if k.startswith("param.")}


def rel_err(a, b):
    """max |a-b| / max |b|  (scale-relative, the metric the north star states)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    den = float(b.abs().max())
    if den == 0.0:
        return float((a - b).abs().max())
    return float((a - b).abs().max()) / den


def greedy_raw_from_table(table):
    """Reconstruct the raw argmax sequence from the eos-swept table of results
    (row e = what decode() returned with eos_id = e; -1 padded)."""
    best = []
    for row in table:
        r = [int(v) for v in row if v >= 0]
        if len(r) > len(best):
            best = r
    return best


class TinyLM(torch.nn.Module):
    """A deterministic stand-in for the language model of shallow fusion (Models/decoder.py:93-96,180-183 call
    ``model_lm(tokens (n, len) int64)[:, -1, :]``): score[b, t] = W tanh(mean of the embeddings of tokens 0..t) + bias,
    causal and row-independent like a real LM.  Used by oracle/make_golden_lm.py (with the reference) and by the tests
    (with the oracle and the HIP path) with the weights stored in tests/golden/lm/*.npz."""

    def __init__(self, V, width=8):
        super().__init__()
        self.emb = torch.nn.Embedding(V, width)
        self.out = torch.nn.Linear(width, V)

    def forward(self, tokens):
        e = self.emb(tokens)                                            # (n, len, width)
        steps = torch.arange(1, tokens.shape[1] + 1, device=e.device, dtype=e.dtype).view(1, -1, 1)
        return self.out(torch.tanh(torch.cumsum(e, 1) / steps))


LM_CASES = ("tiny_a", "tiny_b", "long_t")
LM_RUNS = (("lmgreedy", "decode", 1), ("lmbeam3", "decode", 3), ("lmv2beam3", "decode_v2", 3))


def load_lm(name):
    """tests/golden/lm/<name>.npz -> (npz, TinyLM with its stored weights, lm_weight)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm", name + ".npz"), allow_pickle=False)
    lm = TinyLM(z["lm.emb.weight"].shape[0], z["lm.emb.weight"].shape[1])
    lm.load_state_dict({k[len("lm."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("lm.")})
    return z, lm.eval(), float(z["lm_weight"])
