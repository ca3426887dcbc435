"""CPU: the oracle restatement vs the golden vectors produced by the reference."""
import numpy as np
import pytest
import torch

from oracle import ref_model as R
from tests._util import (CASES, LM_CASES, LM_RUNS, V2_CASES, V2_RUNS, load_case, load_lm, load_v2, params_of, rel_err,
                         greedy_raw_from_table)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-10)])
def test_forward_backward_matches_reference(name, tag, dtype, tol):
    z, hp = load_case(name)
    m = R.AttModel(hp).to(dtype)
    m.load_state_dict(params_of(z, dtype))
    m.train()
    x = torch.from_numpy(z["x"]).to(dtype).requires_grad_(True)
    lengths = torch.from_numpy(z["lengths"])
    targets = torch.from_numpy(z["targets"])
    tl = torch.from_numpy(z["text_lengths"])
    y = m(x, lengths, targets)
    loss = R.label_smoothing_loss(y, targets, tl, hp.T_norm, hp.B_norm)
    loss.backward()
    assert rel_err(y.detach(), z[f"{tag}.youtput"]) < tol
    assert abs(float(loss) - float(z[f"{tag}.loss"])) / abs(float(z[f"{tag}.loss"])) < tol
    assert rel_err(x.grad, z[f"{tag}.dx"]) < 50 * tol
    for k, p in m.named_parameters():
        assert rel_err(p.grad, z[f"{tag}.grad.{k}"]) < 50 * tol, k


@pytest.mark.parametrize("name", CASES)
def test_state_dict_keys_identical(name):
    z, hp = load_case(name)
    m = R.AttModel(hp)
    want = sorted(k[len("param."):] for k in z.files if k.startswith("param."))
    assert sorted(m.state_dict().keys()) == want
    assert len(want) == 4 * 2 * hp.num_encoder_layer + 15


@pytest.mark.parametrize("name", CASES)
def test_closed_form_loss_and_grad(name):
    z, hp = load_case(name)
    y = torch.from_numpy(z["f64.youtput"]).requires_grad_(True)
    targets = torch.from_numpy(z["targets"])
    tl = torch.from_numpy(z["text_lengths"])
    dense = R.label_smoothing_loss(y, targets, tl, hp.T_norm, hp.B_norm)
    dense.backward()
    loss, dz = R.label_smoothing_loss_closed_form(y.detach(), targets, tl, hp.T_norm, hp.B_norm)
    assert abs(float(loss) - float(z["f64.loss"])) < 1e-10 * abs(float(z["f64.loss"]))
    assert rel_err(dz, y.grad) < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_explicit_blstm_equals_packed(name):
    z, hp = load_case(name)
    m = R.AttModel(hp).double()
    m.load_state_dict(params_of(z, torch.float64))
    x = torch.from_numpy(z["x"]).double()
    lengths = torch.from_numpy(z["lengths"])
    with torch.no_grad():
        a = m.encoder(x, lengths)
        b = R.blstm_explicit(x, lengths, m.encoder.bi_lstm)
    assert rel_err(b, a) < 1e-12
    for i, n in enumerate(lengths.tolist()):
        assert float(a[i, n:].abs().max()) == 0.0 if n < x.shape[1] else True


@pytest.mark.parametrize("name", CASES)
def test_greedy_tokens(name):
    z, hp = load_case(name)
    m = R.AttModel(hp)
    m.load_state_dict(params_of(z))
    m.eval()
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(z["greedy.max_len"])
    for utt in range(2):
        if f"greedy.utt{utt}" not in z.files:
            continue
        table = z[f"greedy.utt{utt}"]
        hb = m.encoder(x[utt:utt + 1], lengths[utt:utt + 1])
        _, raw = m.decoder.decode_greedy(hb, lengths[utt:utt + 1], max_len, eos_id=-1)
        assert len(raw) == max_len
        for eos in range(hp.num_classes):
            want = [int(v) for v in table[eos] if v >= 0]
            got = raw[:raw.index(eos) + 1] if eos in raw else []
            assert got == want, (utt, eos)
        assert greedy_raw_from_table(table) == raw[:len(greedy_raw_from_table(table))]


@pytest.mark.parametrize("name", CASES)
def test_beam_tokens(name):
    """The oracle's beam search against what the reference's decode() returned (width 3 with
    the log_softmax score on utterances 0/1, width 2 with the softmax score on utterance 0),
    for every eos id."""
    z, hp = load_case(name)
    m = R.AttModel(hp)
    m.load_state_dict(params_of(z))
    m.eval()
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(z["greedy.max_len"])
    for tag, width, score in (("beam3", 3, "log_softmax"), ("beam2sm", 2, "softmax")):
        for utt in range(2):
            if f"{tag}.utt{utt}" not in z.files:
                continue
            table = z[f"{tag}.utt{utt}"]
            hb = m.encoder(x[utt:utt + 1], lengths[utt:utt + 1])
            for eos in range(hp.num_classes):
                want = [int(v) for v in table[eos] if v >= 0]
                got, _ = m.decoder.decode_beam(hb, lengths[utt:utt + 1], width, max_len, eos, score)
                assert got == want, (tag, utt, eos)


@pytest.mark.parametrize("name", V2_CASES)
def test_beam_v2_tokens(name):
    """The oracle's decode_beam_v2 against what the reference's decode_v2() returned (widths 3 and 4
    with the log_softmax score, 2 and 6 with the softmax score), for every eos id -- including the
    searches whose answer is [] because an empty result slot wins the final argmax."""
    z, hp = load_case(name)
    v2 = load_v2(name)
    m = R.AttModel(hp)
    m.load_state_dict(params_of(z))
    m.eval()
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    max_len = int(v2["max_len"])
    seen = 0
    for tag, width, score in V2_RUNS:
        for utt in range(2):
            if f"{tag}.utt{utt}" not in v2.files:
                continue
            table = v2[f"{tag}.utt{utt}"]
            hb = m.encoder(x[utt:utt + 1], lengths[utt:utt + 1])
            for eos in range(hp.num_classes):
                want = [int(v) for v in table[eos] if v >= 0]
                got, _ = m.decoder.decode_beam_v2(hb, lengths[utt:utt + 1], width, max_len, eos, score)
                assert got == want, (tag, utt, eos)
                seen += bool(want)
    assert seen > 0


@pytest.mark.parametrize("name", LM_CASES)
def test_shallow_fusion_tokens(name):
    """The oracle's decode / decode_v2 with a language model (decoder.py:93-96, :180-183) against what the reference
    returned with the same LM (tests/golden/lm, oracle/make_golden_lm.py): widths 1 and 3 of decode, width 3 of
    decode_v2, log_softmax scoring, lm_weight 0.7, every eos id; and the LM does change some of the answers."""
    z, hp = load_case(name)
    g, lm, lm_weight = load_lm(name)
    m = R.AttModel(hp)
    m.load_state_dict(params_of(z))
    m.eval()
    x = torch.from_numpy(z["x"])
    lengths = torch.from_numpy(z["lengths"])
    hp.max_decoder_seq_len, hp.lm_weight, hp.score_func = int(g["max_len"]), lm_weight, "log_softmax"
    seen = changed = 0
    for tag, fn, width in LM_RUNS:
        hp.beam_width = width
        for utt in range(2):
            table = g[f"{tag}.utt{utt}"]
            for eos in range(hp.num_classes):
                hp.eos_id = eos
                want = [int(v) for v in table[eos] if v >= 0]
                got = getattr(m, fn)(x[utt:utt + 1], lengths[utt:utt + 1], lm, None)
                assert got == want, (tag, utt, eos)
                seen += bool(want)
                changed += got != getattr(m, fn)(x[utt:utt + 1], lengths[utt:utt + 1], None, None)
    assert seen > 0 and changed > 0
    hp.score_func = "softmax"           # broken in the reference (decoder.py:101-106, :188-193): refused
    with pytest.raises(ValueError):
        m.decode(x[:1], lengths[:1], lm, None)


def test_beam_v2_rejects_unequal_widths():
    """decoder.py:219,241: the reference raises when 2 * decoder width != encoder output width."""
    z, hp = load_case("mixed_h")
    m = R.AttModel(hp)
    m.load_state_dict(params_of(z))
    hb = m.encoder(torch.from_numpy(z["x"])[:1], torch.from_numpy(z["lengths"])[:1])
    with pytest.raises(RuntimeError):
        m.decoder.decode_beam_v2(hb, torch.from_numpy(z["lengths"])[:1], 3, 5, 1)
