"""CPU oracle for the AttModel training hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-CPU ops, the algorithm of the reference's
hot path.  It is imported only by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, as the checker / the timed CPU comparator.
The product path (``asr_amd``) never imports it.

Reference lines followed (paths relative to /root/reference):

* frame stacking ........ utils/utils.py:47-65
* BLSTM encoder ......... Models/encoder.py:25-35   (torch.nn.LSTM over a packed
                          batch; third-party arithmetic = the documented PyTorch
                          LSTM equations, gate order i,f,g,o, b_ih + b_hh)
* location attention .... Models/attention.py:24-40
* decoder loop + cell ... Models/decoder.py:36-61, 269-279
* greedy decode ......... Models/decoder.py:63-130  (beam_width == 1)
* beam searches ......... Models/decoder.py:63-130 (beam_width > 1), :132-267 (decode_v2), with the shallow-fusion
                          lines :93-96 / :180-183 (pinned by tests/golden/v2 and tests/golden/lm)
* label smoothing ....... Loss/label_smoothing.py:14-28

Parity pin: ``tests/golden/*.npz`` were produced by importing the reference
itself (``oracle/make_golden.py``); ``tests/test_oracle_golden.py`` checks this
restatement against every one of them.

Module/parameter names are kept identical to the reference (31 tensors,
SURVEY.md section 8b) so a ``state_dict`` moves freely between the reference,
this oracle and ``asr_amd.Models.AttModel``.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# utils/utils.py:47-65
# --------------------------------------------------------------------------
def frame_stacking(x, x_lengths, stack):
    if stack == 1:
        return x, x_lengths
    n = x.shape[1] // stack
    return x[:, : n * stack].reshape(x.shape[0], n, -1), x_lengths // stack


# --------------------------------------------------------------------------
# Models/encoder.py:14-35
# --------------------------------------------------------------------------
class Encoder(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.bi_lstm = nn.LSTM(
            input_size=hp.lmfb_dim * hp.frame_stacking,
            hidden_size=hp.num_hidden_nodes_encoder,
            num_layers=hp.num_encoder_layer,
            batch_first=True,
            dropout=hp.encoder_dropout,
            bidirectional=True,
        )

    def forward(self, x, lengths):
        T = x.shape[1]
        packed = nn.utils.rnn.pack_padded_sequence(
            x, lengths.cpu(), batch_first=True, enforce_sorted=False)
        h, _ = self.bi_lstm(packed)
        hbatch, _ = nn.utils.rnn.pad_packed_sequence(h, batch_first=True, total_length=T)
        return hbatch


def blstm_explicit(x, lengths, lstm: nn.LSTM):
    """Loop restatement of the packed BLSTM (no dropout), for small shapes.

    Spells out what ``nn.LSTM`` over a packed batch computes: every sequence
    runs over its own ``lengths[b]`` frames (the reverse direction starts at
    frame ``lengths[b]-1`` from a zero state) and rows past the length are 0
    (Models/encoder.py:32-34).  Used by the tests to pin the masking rules the
    HIP recurrence kernels implement without packing.
    """
    B, T, _ = x.shape
    H = lstm.hidden_size
    inp = x
    for layer in range(lstm.num_layers):
        outs = []
        for d, sfx in enumerate(("", "_reverse")):
            w_ih = getattr(lstm, f"weight_ih_l{layer}{sfx}")
            w_hh = getattr(lstm, f"weight_hh_l{layer}{sfx}")
            b = getattr(lstm, f"bias_ih_l{layer}{sfx}") + getattr(lstm, f"bias_hh_l{layer}{sfx}")
            h = x.new_zeros(B, H)
            c = x.new_zeros(B, H)
            out = [None] * T
            order = range(T) if d == 0 else range(T - 1, -1, -1)
            for t in order:
                gates = inp[:, t] @ w_ih.t() + h @ w_hh.t() + b
                i, f, g, o = gates.chunk(4, 1)
                c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
                h_new = torch.sigmoid(o) * torch.tanh(c_new)
                valid = (t < lengths).to(x.dtype).unsqueeze(1).to(x.device)
                c = valid * c_new + (1 - valid) * c
                h = valid * h_new + (1 - valid) * h
                out[t] = valid * h_new
            outs.append(torch.stack(out, 1))
        inp = torch.cat(outs, 2)
    return inp


# --------------------------------------------------------------------------
# Models/attention.py:6-40
# --------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, hp):
        super().__init__()
        hd = hp.num_hidden_nodes_decoder
        he = hp.num_hidden_nodes_encoder * (1 if getattr(hp, "encoder_type", None) == "Conformer" else 2)   # attention.py:13-16, decoder.py:19-22
        self.L_se = nn.Linear(hd, hd * 2, bias=False)
        self.L_he = nn.Linear(he, hd * 2)
        self.L_ee = nn.Linear(hd * 2, 1, bias=False)
        self.L_fe = nn.Linear(10, hd * 2, bias=False)
        self.F_conv1d = nn.Conv1d(1, 10, 100, stride=1, padding=50, bias=False)

    def forward(self, s, hbatch, alpha, e_mask):
        T = hbatch.size(1)
        f = self.F_conv1d(alpha).transpose(1, 2)[:, :T, :]          # (B,T,10)
        e = torch.tanh(self.L_se(s).unsqueeze(1) + self.L_he(hbatch) + self.L_fe(f))
        e = self.L_ee(e)                                            # (B,T,1)
        w = (e - e.max(1)[0].unsqueeze(1)).exp() * e_mask           # max over ALL T
        alpha = w / w.sum(dim=1, keepdim=True)
        g = (alpha * hbatch).sum(dim=1)
        return g, alpha.transpose(1, 2)


def lstm_cell_tanh_form(x, c):
    """Models/decoder.py:269-279 -- sigmoid written as 0.5*tanh(0.5x)+0.5."""
    i, f, g, o = x.chunk(4, 1)
    i = torch.tanh(i * 0.5) * 0.5 + 0.5
    f = torch.tanh(f * 0.5) * 0.5 + 0.5
    g = torch.tanh(g)
    o = torch.tanh(o * 0.5) * 0.5 + 0.5
    c_next = f * c + i * g
    return o * torch.tanh(c_next), c_next


# --------------------------------------------------------------------------
# Models/decoder.py:14-61 (+ greedy subset of :63-130)
# --------------------------------------------------------------------------
class Decoder(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        hd = hp.num_hidden_nodes_decoder
        he = hp.num_hidden_nodes_encoder * (1 if getattr(hp, "encoder_type", None) == "Conformer" else 2)   # attention.py:13-16, decoder.py:19-22
        self.hd = hd
        self.att = Attention(hp)
        self.L_sy = nn.Linear(hd, hd, bias=False)
        self.L_gy = nn.Linear(he, hd)
        self.L_yy = nn.Linear(hd, hp.num_classes)
        self.L_ys = nn.Embedding(hp.num_classes, hd * 4)
        self.L_ss = nn.Linear(hd, hd * 4, bias=False)
        self.L_gs = nn.Linear(he, hd * 4)

    @staticmethod
    def _mask(hbatch, lengths):
        B, T, _ = hbatch.shape
        t = torch.arange(T, device=hbatch.device).unsqueeze(0)
        return (t < lengths.to(hbatch.device).unsqueeze(1)).to(hbatch.dtype).unsqueeze(2)

    def forward(self, hbatch, lengths, targets):
        B, T, _ = hbatch.shape
        L = targets.size(1)
        e_mask = self._mask(hbatch, lengths)
        s = hbatch.new_zeros(B, self.hd)
        c = hbatch.new_zeros(B, self.hd)
        alpha = hbatch.new_zeros(B, 1, T)
        ys = []
        for step in range(L):
            g, alpha = self.att(s, hbatch, alpha, e_mask)
            ys.append(self.L_yy(torch.tanh(self.L_gy(g) + self.L_sy(s))))
            rec = self.L_ys(targets[:, step]) + self.L_ss(s) + self.L_gs(g)
            s, c = lstm_cell_tanh_form(rec, c)
        return torch.stack(ys, 1)

    @torch.no_grad()
    def decode_greedy(self, hbatch, lengths, max_len, eos_id):
        """beam_width==1 path of Models/decoder.py:63-130 (batch 1).

        Returns (result, raw): ``result`` is what the reference returns (the
        sequence if it ended in eos, else []), ``raw`` the tokens emitted."""
        B, T, _ = hbatch.shape
        e_mask = self._mask(hbatch, lengths)
        s = hbatch.new_zeros(B, self.hd)
        c = hbatch.new_zeros(B, self.hd)
        alpha = hbatch.new_zeros(B, 1, T)
        seq = []
        for _ in range(max_len):
            g, alpha = self.att(s, hbatch, alpha, e_mask)
            y = F.log_softmax(self.L_yy(torch.tanh(self.L_gy(g) + self.L_sy(s))), dim=1)
            best = y.argmax(1)
            seq.append(int(best[0]))
            rec = self.L_ys(best) + self.L_ss(s) + self.L_gs(g)
            s, c = lstm_cell_tanh_form(rec, c)
            if seq[-1] == eos_id:
                return list(seq), seq
        return [], seq


    @torch.no_grad()
    def decode_beam(self, hbatch, lengths, beam_width, max_len, eos_id, score_func="log_softmax", model_lm=None,
                    lm_weight=0.0):
        """Models/decoder.py:63-130 with beam_width >= 1 (batch 1).  Shallow fusion (:93-96): a hypothesis that already
        holds tokens adds ``lm_weight * log_softmax(model_lm(its tokens)[:, -1])`` to its log-probabilities before its
        best tokens are picked; the first step has no LM term.  With score_func == 'softmax' the reference's LM branch
        (:101-104) never assigns the scores it then reads: not restated (ValueError).

        Every live hypothesis is advanced by one attention step, its ``beam_width`` best tokens
        (repeated argmax, :109-116) become candidates scored ``old score + (log-)probability``
        (:115; with score_func == 'softmax' the reference adds probabilities), all candidates are
        sorted by score with a stable sort (:122) and the best ``beam_width`` survive.  The search
        stops as soon as the best survivor ends in eos (:125-128) and returns that sequence;
        otherwise [] after ``max_len`` steps.  Returns (result, trace): trace[k] = the survivors of
        step k as (tokens, score)."""
        assert hbatch.size(0) == 1
        T = hbatch.size(1)
        e_mask = self._mask(hbatch, lengths)
        beams = [([], 0.0, (hbatch.new_zeros(1, self.hd), hbatch.new_zeros(1, self.hd), hbatch.new_zeros(1, 1, T)))]
        trace = []
        for _ in range(max_len):
            cands = []
            for seq, score, (c, s, alpha) in beams:
                g, alpha_n = self.att(s, hbatch, alpha, e_mask)
                y = self.L_yy(torch.tanh(self.L_gy(g) + self.L_sy(s)))
                y = F.log_softmax(y, dim=1) if score_func == "log_softmax" else F.softmax(y, dim=1)
                tmpy = y.clone()
                if model_lm is not None:
                    if score_func != "log_softmax":
                        raise ValueError("shallow fusion is defined for score_func == 'log_softmax' only (decoder.py:101-106)")
                    if len(seq) > 0:
                        lm_score = model_lm(torch.tensor([seq], dtype=torch.int64))[:, -1, :]
                        tmpy = y + lm_weight * F.log_softmax(lm_score.to(y.dtype), dim=1)
                for _k in range(beam_width):
                    best = tmpy.argmax(1)
                    cand_score = score + float(tmpy[0, best])
                    tmpy[0, best] = -10000000000.0
                    rec = self.L_ys(best) + self.L_ss(s) + self.L_gs(g)
                    s_n, c_n = lstm_cell_tanh_form(rec, c)
                    cands.append((seq + [int(best)], cand_score, (c_n, s_n, alpha_n)))
            cands.sort(key=lambda t: t[1], reverse=True)      # stable, like sorted(..., reverse=True)
            beams = cands[:beam_width]
            trace.append([(list(b[0]), float(b[1])) for b in beams])
            if beams[0][0][-1] == eos_id:
                return list(beams[0][0]), trace
        return [], trace

    @torch.no_grad()
    def decode_beam_v2(self, hbatch, lengths, beam_width, max_len, eos_id, score_func="log_softmax", model_lm=None,
                       lm_weight=0.0):
        """Models/decoder.py:132-267 (decode_v2, batch 1): the second beam search.  Shallow fusion (:180-183): from the
        second step on, ``lm_weight * log_softmax(model_lm(live sequences (W, step))[:, -1])`` is added to the rows'
        log-probabilities before the top-k; 'softmax' scoring with an LM is broken in the reference (:188-193) and
        not restated (ValueError).

        The ``beam_width`` live hypotheses are the batch rows of one attention step (:172-176; all
        rows start from the zero state, :146-151).  Each row proposes its ``beam_width`` best tokens
        (:195), scored ``old + (log-)probability + 1`` in float32 (:196: a bonus of 1 per token).
        Step 0 keeps row 0's proposals as the first beam without looking at eos (:200-210).  Later
        steps walk the best ``2 * beam_width`` of all proposals in score order (:212-214): one that
        ends in eos is moved to the next free result slot (:222-230), any other fills the next beam
        slot with its parent's state, context and attention (:231-244), until the beam is full or
        ``beam_width`` results exist, which also ends the search (:246-255).  The answer is the
        result slot with the highest score (:256-258) -- empty slots hold score 0 and length 0, so
        [] comes back whenever every finished hypothesis scores below 0 (or none finished).

        Raises RuntimeError as the reference does when 2 * decoder width != encoder output width
        (the contexts are gathered into a buffer of that width, :219,241).
        Returns (result, trace): trace[k] = (live survivors of step k as (tokens, score), results
        finished so far as (tokens, score))."""
        assert hbatch.size(0) == 1
        W = int(beam_width)
        if hbatch.size(2) != 2 * self.hd:
            raise RuntimeError("decode_v2: encoder output width must be 2 * num_hidden_nodes_decoder")
        T = hbatch.size(1)
        e_mask = self._mask(hbatch, lengths)
        s = hbatch.new_zeros(W, self.hd)
        c = hbatch.new_zeros(W, self.hd)
        alpha = hbatch.new_zeros(W, 1, T)
        score = hbatch.new_zeros(W)
        seqs = [[] for _ in range(W)]
        done = []                                                   # (tokens, score), slot order
        trace = []
        for step in range(max_len):
            g, alpha_n = self.att(s, hbatch, alpha, e_mask)
            y = self.L_yy(torch.tanh(self.L_gy(g) + self.L_sy(s)))
            y = F.log_softmax(y, dim=1) if score_func == "log_softmax" else F.softmax(y, dim=1)
            if model_lm is not None:
                if score_func != "log_softmax":
                    raise ValueError("shallow fusion is defined for score_func == 'log_softmax' only (decoder.py:188-193)")
                if step > 0:
                    lm_score = model_lm(torch.tensor(seqs, dtype=torch.int64))[:, -1, :]
                    y = y + lm_weight * F.log_softmax(lm_score.to(y.dtype), dim=1)
            best, idx = y.topk(W, dim=1)
            cand = score.unsqueeze(1) + best + 1
            if step == 0:
                # all rows are identical here; row i keeps its own (zero) state
                parents, toks, new_score = list(range(W)), idx[0].tolist(), cand[0].clone()
            else:
                k_scores, k_ix = cand.reshape(-1).topk(2 * W)
                parents, toks, kept = [], [], []
                for sc, ix in zip(k_scores.tolist(), k_ix.tolist()):
                    h, j = ix // W, ix % W
                    tok = int(idx[h, j])
                    if tok == eos_id:
                        done.append((seqs[h] + [tok], sc))
                    else:
                        parents.append(h)
                        toks.append(tok)
                        kept.append(sc)
                    if len(parents) == W or len(done) >= W:
                        break
                if len(done) >= W:
                    trace.append(([], list(done)))
                    break
                new_score = torch.tensor(kept, dtype=score.dtype)
            sel = torch.tensor(parents)
            rec = self.L_ys(torch.tensor(toks)) + self.L_ss(s[sel]) + self.L_gs(g[sel])
            s, c = lstm_cell_tanh_form(rec, c[sel])
            alpha = alpha_n[sel]
            seqs = [seqs[h] + [t] for h, t in zip(parents, toks)]
            score = new_score
            trace.append(([(list(q), float(v)) for q, v in zip(seqs, score)], list(done)))
        slots = [0.0] * W
        for i, (_, sc) in enumerate(done[:W]):
            slots[i] = sc
        best_slot = max(range(W), key=lambda i: (slots[i], -i))    # argmax, first on ties
        result = list(done[best_slot][0]) if best_slot < len(done) else []
        return result, trace


class AttModel(nn.Module):
    """Models/AttModel.py:10-39, BLSTM branch."""

    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.encoder = Encoder(hp)
        self.decoder = Decoder(hp)

    def forward(self, x, lengths, targets, src_pos=None):
        return self.decoder(self.encoder(x, lengths), lengths, targets)

    def decode(self, x, lengths, model_lm=None, src_pos=None):
        with torch.no_grad():
            hbatch = self.encoder(x, lengths)
            if getattr(self.hp, "beam_width", 1) > 1 or model_lm is not None:
                return self.decoder.decode_beam(hbatch, lengths, self.hp.beam_width, self.hp.max_decoder_seq_len,
                                                self.hp.eos_id, getattr(self.hp, "score_func", "log_softmax"),
                                                model_lm, getattr(self.hp, "lm_weight", 0.0))[0]
            return self.decoder.decode_greedy(
                hbatch, lengths, self.hp.max_decoder_seq_len, self.hp.eos_id)[0]

    def decode_v2(self, x, lengths, model_lm=None, src_pos=None):
        """Models/AttModel.py:41-49."""
        with torch.no_grad():
            hbatch = self.encoder(x, lengths)
            return self.decoder.decode_beam_v2(hbatch, lengths, self.hp.beam_width, self.hp.max_decoder_seq_len,
                                               self.hp.eos_id, getattr(self.hp, "score_func", "log_softmax"),
                                               model_lm, getattr(self.hp, "lm_weight", 0.0))[0]


# --------------------------------------------------------------------------
# Loss/label_smoothing.py:14-28
# --------------------------------------------------------------------------
def label_smoothing_loss(predicted_label, target, text_lengths, T_norm=True, B_norm=False, eps=0.1):
    """Dense restatement (same op sequence as the reference, minus ``.cuda()``)."""
    B, L, V = predicted_label.shape
    logp = F.log_softmax(predicted_label, dim=2)
    q = torch.zeros(B * L, V, dtype=logp.dtype, device=logp.device).scatter(1, target.reshape(-1, 1), 1)
    q = (q * (1 - eps) + (1 - q) * eps / (V - 1)).view(B, L, V)
    loss = 0.0
    for i, t in enumerate(text_lengths.tolist()):
        term = -(q[i, :t] * logp[i, :t]).sum()
        loss = loss + (term / t if T_norm else term)
    if B_norm:
        loss = loss / B
    return loss


def label_smoothing_loss_closed_form(z, target, text_lengths, T_norm=True, B_norm=False, eps=0.1):
    """Row-wise closed form of the same loss and its gradient (what the HIP kernel does).

    loss_row = -(1-eps) lp_y - eps/(V-1) (sum_v z_v - V lse - lp_y)
    dz       = w_row (softmax(z) - q),   w_row = 1/t_i (T_norm) for t < t_i else 0
    """
    B, L, V = z.shape
    lse = torch.logsumexp(z, dim=2)
    zy = z.gather(2, target.unsqueeze(2)).squeeze(2)
    lpy = zy - lse
    off = eps / (V - 1)
    row = -(1 - eps) * lpy - off * (z.sum(2) - V * lse - lpy)
    steps = torch.arange(L, device=z.device).unsqueeze(0)
    tl = text_lengths.to(z.device).unsqueeze(1)
    w = (steps < tl).to(z.dtype)
    if T_norm:
        w = w / tl.to(z.dtype)
    if B_norm:
        w = w / B
    loss = (row * w).sum()
    q = torch.full_like(z, off).scatter(2, target.unsqueeze(2), 1 - eps)
    dz = w.unsqueeze(2) * (torch.softmax(z, 2) - q)
    return loss, dz


def make_hp(**kw):
    """hp attribute bag with the template defaults (configs/hparams_template.py,
    utils/utils.py:199-220) for the attributes the hot path reads."""
    import types
    d = dict(lmfb_dim=80, frame_stacking=3, num_hidden_nodes=320, num_encoder_layer=5,
             encoder_dropout=0.2, encoder_type=None, decoder_type='Attention',
             num_classes=1000, eos_id=1, T_norm=True, B_norm=False, beam_width=1,
             max_decoder_seq_len=200, score_func='log_softmax', clip=5.0, accum_grad=1,
             lr=1e-3, batch_size=32)
    d.update(kw)
    d.setdefault('num_hidden_nodes_encoder', d['num_hidden_nodes'])
    d.setdefault('num_hidden_nodes_decoder', d['num_hidden_nodes'])
    return types.SimpleNamespace(**d)
