"""CPU oracle for the CTC head.  TEST INFRASTRUCTURE ONLY.

Imported only by ``tests/`` (and never by ``asr_amd``).  Restates, in plain torch-CPU ops:

* CTCModel .............. Models/CTCModel.py:8-23   (encoder + nn.Linear(2*num_hidden_nodes, V))
* greedy CTC decode ..... Models/CTCModel.py:25-42  (collapse repeats, drop id 0; ``prev_id`` is
                          NOT reset between utterances and every padded frame is read -- kept)
* the training loss ..... train.py:56-57            (F.log_softmax(.., 2).transpose(0, 1), then
                          F.ctc_loss(.., blank=0): reduction 'mean', zero_infinity False)

Third-party arithmetic: the CTC recursion itself is torch's (``F.ctc_loss``), not the
reference's; ``ctc_nll_explicit`` below spells out the published algorithm (Graves et al. 2006,
log-space alpha recursion over the blank-augmented path) in float64 loops and the tests hold
``F.ctc_loss`` to it.

Parity pin: ``tests/golden/ctc/*.npz`` were produced by importing the reference's CTCModel and
running train.py:56-57 on its output (``oracle/make_golden_ctc.py``);
``tests/test_oracle_golden.py`` checks this restatement against every one of them.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_model import Encoder


class CTCModel(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.encoder = Encoder(hp)
        self.decoder = nn.Linear(hp.num_hidden_nodes * 2, hp.num_classes)

    def forward(self, x, lengths, targets=None, pos_mel=None):
        return self.decoder(self.encoder(x, lengths))

    def decode(self, x, lengths, model_lm=None, src_pos=None):
        with torch.no_grad():
            out = self.forward(x, lengths)
        results = []
        prev_id = self.hp.num_classes + 1
        for b in range(out.shape[0]):
            res = []
            for v in out[b].argmax(dim=1):
                if int(v) != prev_id and int(v) != 0:
                    res.append(int(v))
                prev_id = int(v)
            results.append(res)
        return results


def ctc_loss_train(predict_ts, text, mel_lengths, text_lengths):
    """train.py:56-57."""
    lp = F.log_softmax(predict_ts, dim=2).transpose(0, 1)
    return F.ctc_loss(lp, text, mel_lengths, text_lengths, blank=0)


def ctc_nll_explicit(logits, targets, in_len, tgt_len, blank=0):
    """Per-utterance nll by the textbook recursion, float64 python loops (small cases)."""
    lp = torch.log_softmax(logits.double(), dim=-1)
    out = []
    for b in range(logits.shape[0]):
        T, L = int(in_len[b]), int(tgt_len[b])
        path = [blank]
        for l in targets[b, :L].tolist():
            path += [int(l), blank]
        S = len(path)
        ninf = -math.inf
        if T == 0:
            out.append(0.0 if L == 0 else math.inf)
            continue
        alpha = [ninf] * S
        alpha[0] = float(lp[b, 0, path[0]])
        if S > 1:
            alpha[1] = float(lp[b, 0, path[1]])
        for t in range(1, T):
            new = [ninf] * S
            for s in range(S):
                terms = [alpha[s]]
                if s >= 1:
                    terms.append(alpha[s - 1])
                if s >= 2 and path[s] != path[s - 2]:
                    terms.append(alpha[s - 2])
                m = max(terms)
                if m == ninf:
                    continue
                new[s] = m + math.log(sum(math.exp(v - m) for v in terms)) + float(lp[b, t, path[s]])
            alpha = new
        terms = [alpha[S - 1]] + ([alpha[S - 2]] if S > 1 else [])
        m = max(terms)
        out.append(math.inf if m == ninf else -(m + math.log(sum(math.exp(v - m) for v in terms))))
    return torch.tensor(out, dtype=torch.float64)
