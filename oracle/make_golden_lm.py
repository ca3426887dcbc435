"""Generate tests/golden/lm/<case>.npz: what THE REFERENCE's decode / decode_v2 return with a language model
(shallow fusion, Models/decoder.py:93-96 and :180-183; this container only).

    python oracle/make_golden_lm.py

For three cases of tests/golden/*.npz (weights and inputs read back from the fixture) the reference's
``AttModel.decode`` (beam width 1 and 3) and ``AttModel.decode_v2`` (width 3) run unchanged with
``score_func = 'log_softmax'``, ``hp.lm_weight = 0.7`` and ``model_lm`` = tests/_util.py::TinyLM (a small causal
stand-in whose seeded weights are stored in the fixture: the reference only ever calls ``model_lm(tokens)[:, -1, :]``),
the eos id swept over the vocabulary.  The same sweep without the LM is already in tests/golden/*.npz and
tests/golden/v2/*.npz; ``changed`` counts the eos ids whose answer the LM changes.  'softmax' scoring with an LM is
broken in the reference itself (decoder.py:101-106 reads ``tmpy`` without assigning it) and has no vectors.
decode_v2's trailing sentencepiece load (decoder.py:261-264) gets a throw-away model, as in make_golden_v2.py.
Only data is committed.
"""
import ast
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle.make_golden_v2 import throwaway_spm          # noqa: E402
from tests._util import LM_RUNS, TinyLM                    # noqa: E402

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
GOLDEN = os.path.join(HERE, "..", "tests", "golden")
CASES = ("tiny_a", "tiny_b", "long_t")
MAX_LEN, LM_WEIGHT = 16, 0.7


def main():
    sys.path.insert(0, REF)
    from Models.AttModel import AttModel        # noqa: E402  (the reference)
    torch.set_num_threads(4)
    os.makedirs(os.path.join(GOLDEN, "lm"), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        spm_model = throwaway_spm(tmp)
        for name in CASES:
            z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
            hp = types.SimpleNamespace(**dict(ast.literal_eval(str(z["meta_hp"]))))
            hp.spm_model, hp.lm_weight, hp.score_func, hp.max_decoder_seq_len = spm_model, LM_WEIGHT, "log_softmax", MAX_LEN
            model = AttModel(hp)
            model.load_state_dict({k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
            model.eval()
            V = hp.num_classes
            torch.manual_seed(1000 + V)
            lm = TinyLM(V).eval()
            with torch.no_grad():
                lm.out.weight.mul_(6.0)          # peaked enough to overturn the acoustic model's choices
            x, lens = torch.from_numpy(z["x"]), torch.from_numpy(z["lengths"])
            out = {"max_len": np.array(MAX_LEN), "lm_weight": np.array(LM_WEIGHT)}
            for k, v in lm.state_dict().items():
                out["lm." + k] = v.numpy().copy()
            changed = {}
            for tag, fn, width in LM_RUNS:
                hp.beam_width = width
                for utt in range(min(2, x.shape[0])):
                    res = -np.ones((V, MAX_LEN), dtype=np.int64)
                    diff = 0
                    for eos in range(V):
                        hp.eos_id = eos
                        with torch.no_grad():
                            r = getattr(model, fn)(x[utt:utt + 1], lens[utt:utt + 1], lm, None)
                            r0 = getattr(model, fn)(x[utt:utt + 1], lens[utt:utt + 1], None, None)
                        res[eos, :len(r)] = r
                        diff += list(r) != list(r0)
                    out[f"{tag}.utt{utt}"] = res
                    changed[f"{tag}.utt{utt}"] = diff
            dst = os.path.join(GOLDEN, "lm", name + ".npz")
            np.savez_compressed(dst, **out)
            print(f"{name}: V = {V}; eos ids whose answer the LM changes {changed} -> {os.path.getsize(dst)} bytes")


if __name__ == "__main__":
    main()
