"""Generate tests/golden/v2/<case>.npz: what THE REFERENCE's decode_v2 returns (this container only).

    python oracle/make_golden_v2.py

For every case of tests/golden/*.npz (weights and inputs are read back from the fixture, so the
two sets of vectors describe the same model) the reference's ``AttModel.decode_v2``
(Models/AttModel.py:41-49 -> Models/decoder.py:132-267) is run unchanged on utterances 0 and 1
with beam width 3 / log_softmax and on utterance 0 with width 2 / softmax and width 4 /
log_softmax, on utterances 0 and 1 with width 6 / softmax, the eos id swept over the whole vocabulary (the search returns the best *finished*
hypothesis, so each eos id exposes a different slice of the search).  Only data is committed.

decode_v2 loads ``hp.spm_model`` with sentencepiece after the search (decoder.py:261-264; the
loaded processor is not used): a throw-away model is trained in a temporary directory for that
load.  No LM (model_lm=None).  max_decoder_seq_len is raised to 24 so that more searches fill
all their result slots (an unfilled slot keeps score 0 and wins the final argmax, decoder.py:256,
whenever the finished hypotheses score below 0: the result is then []).  The reference raises for
models whose encoder and decoder widths differ (decoder.py:219,241): no vectors for ``mixed_h``.
"""
import ast
import glob
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = os.environ.get("ASR_REFERENCE", "/root/reference")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
MAX_LEN = 24
RUNS = (("v2beam3", 3, "log_softmax", (0, 1)), ("v2beam2sm", 2, "softmax", (0,)), ("v2beam4", 4, "log_softmax", (0,)),
        ("v2beam6sm", 6, "softmax", (0, 1)))


def throwaway_spm(tmp):
    import sentencepiece as spm
    txt = os.path.join(tmp, "corpus.txt")
    with open(txt, "w") as f:
        for i in range(200):
            f.write(" ".join(f"w{(i * 7 + k) % 23}x" for k in range(6)) + "\n")
    spm.SentencePieceTrainer.Train(f"--input={txt} --model_prefix={os.path.join(tmp, 'm')} --vocab_size=24 "
                                   "--model_type=bpe --minloglevel=2")
    return os.path.join(tmp, "m.model")


def main():
    sys.path.insert(0, REF)
    from Models.AttModel import AttModel        # noqa: E402  (the reference)
    torch.set_num_threads(4)
    os.makedirs(os.path.join(GOLDEN, "v2"), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        spm_model = throwaway_spm(tmp)
        for path in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
            name = os.path.splitext(os.path.basename(path))[0]
            z = np.load(path, allow_pickle=False)
            hp = types.SimpleNamespace(**dict(ast.literal_eval(str(z["meta_hp"]))))
            hp.spm_model = spm_model
            model = AttModel(hp)
            model.load_state_dict({k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")})
            model.eval()
            x = torch.from_numpy(z["x"])
            lens = torch.from_numpy(z["lengths"])
            V = hp.num_classes
            hp.max_decoder_seq_len = MAX_LEN
            if hp.num_hidden_nodes_encoder != hp.num_hidden_nodes_decoder:
                # decoder.py:219,241: the survivors' contexts are gathered into a buffer of width
                # 2 * decoder hidden, so decode_v2 raises when the two widths differ
                hp.beam_width, hp.score_func, hp.eos_id = 3, "log_softmax", 1
                try:
                    model.decode_v2(x[0:1], lens[0:1], None, None)
                    raise SystemExit(f"{name}: expected the reference to raise")
                except RuntimeError as e:
                    print(f"{name}: the reference raises RuntimeError ({str(e)[:60]}...), no vectors")
                continue
            out = {"max_len": np.array(hp.max_decoder_seq_len)}
            for tag, width, score, utts in RUNS:
                hp.beam_width, hp.score_func = width, score
                for utt in utts:
                    if utt >= x.shape[0]:
                        continue
                    res = -np.ones((V, hp.max_decoder_seq_len), dtype=np.int64)
                    for eos in range(V):
                        hp.eos_id = eos
                        r = model.decode_v2(x[utt:utt + 1], lens[utt:utt + 1], None, None)
                        res[eos, :len(r)] = r
                    out[f"{tag}.utt{utt}"] = res
            dst = os.path.join(GOLDEN, "v2", name + ".npz")
            np.savez_compressed(dst, **out)
            n_nonempty = {k: int((v[:, 0] >= 0).sum()) for k, v in out.items() if k != "max_len"}
            print(f"{name}: non-empty results per run {n_nonempty} -> {os.path.getsize(dst)} bytes")


if __name__ == "__main__":
    main()
