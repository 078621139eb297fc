"""Independent pin of the oracle's logit-rule masks (row a7): `transformers`' SuppressTokens / SuppressTokensAtBegin /
WhisperTimeStampLogitsProcessor implement the published Whisper decoding rules the Swift port was written from
(WhisperDecoding.swift:186-290 cites "Python lines ...").  The masks must agree on every history EXCEPT where the Swift port itself
departs from that algorithm -- two spots, both restated faithfully by the oracle and asserted here as exact differences:
  (1) its monotonicity filter takes generated tokens `> timestampBegin` (strict, :254-256), so a generated <|0.00|> is not a "last timestamp";
  (2) it adds the +1 (non-zero segment length) iff penultimateWasTimestamp (:262-264); the published rule adds it unless the history ends
      [text, timestamp] -- they differ exactly when the last TWO generated tokens are text and an earlier timestamp exists.
The timestamp-probability heuristic (:299-322) is switched off in the HF processor here: the port evaluates it on RAW logits, the
published code on masked log-probabilities (a third, documented, divergence)."""
import numpy as np
import pytest
import torch

from oracle import whisper as OW

transformers = pytest.importorskip("transformers")
from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,  # noqa: E402
                                                     WhisperTimeStampLogitsProcessor)

V = 51865          # the multilingual vocabulary: SpecialTokens.for_vocab is the real id arithmetic (WhisperTokenizer.swift:72-96)


class _Cfg:        # the attributes WhisperTimeStampLogitsProcessor reads
    def __init__(self, st, max_initial):
        self.no_timestamps_token_id = st.no_timestamps
        self.eos_token_id = st.eot
        self.bos_token_id = st.eot
        self.max_initial_timestamp_index = max_initial
        self._detect_timestamp_from_logprob = False


def _hf_mask(st, o, init, gen):
    ids = torch.tensor([init + gen], dtype=torch.long)
    scores = torch.zeros(1, V)
    scores = SuppressTokensLogitsProcessor(list(o.suppress_ids))(ids, scores)
    scores = SuppressTokensAtBeginLogitsProcessor(list(o.blank_ids) + [st.eot], begin_index=len(init))(ids, scores)
    scores = WhisperTimeStampLogitsProcessor(_Cfg(st, o.max_initial_timestamp_index), begin_index=len(init),
                                             _detect_timestamp_from_logprob=False)(ids, scores)
    return torch.isinf(scores[0]).numpy()


def _oracle_mask(st, o, init, gen):
    base, ts = OW.rule_masks(init + gen, len(init), st, o, V)
    return torch.isinf(torch.minimum(base, ts)).numpy()


def _setup():
    st = OW.SpecialTokens.for_vocab(V)
    o = OW.DecodingOptions(suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220])
    init, _ = OW.initial_tokens(st, o)
    assert st.no_timestamps + 1 == st.timestamp_begin          # what the HF processor assumes
    return st, o, init


def _histories(st, rng, n):
    """Random generated-token histories that respect nothing in particular: text ids, timestamp ids, in any order."""
    out = [[]]
    for _ in range(n):
        L = int(rng.integers(1, 9))
        h = [int(rng.integers(st.timestamp_begin, st.timestamp_begin + 40)) if rng.random() < 0.45 else int(rng.integers(0, st.eot)) for _ in range(L)]
        out.append(h)
    return out


def test_rule_masks_match_transformers_outside_the_ports_two_departures():
    st, o, init = _setup()
    tsb = st.timestamp_begin
    rng = np.random.default_rng(0)
    n_same = n_strict = n_plus1 = 0
    for gen in _histories(st, rng, 1500):
        hf, ora = _hf_mask(st, o, init, gen), _oracle_mask(st, o, init, gen)
        ts_ge = [t for t in gen if t >= tsb]
        ts_gt = [t for t in gen if t > tsb]
        two_text_tail = len(gen) >= 2 and gen[-1] < tsb and gen[-2] < tsb
        if ts_ge and ts_ge[-1] == tsb and (not ts_gt or ts_gt[-1] != ts_ge[-1]):
            n_strict += 1                                      # departure (1): the last timestamp is <|0.00|> itself
            continue
        if two_text_tail and ts_gt:
            # departure (2): the published rule also forbids repeating the last timestamp; the port allows it
            diff = np.flatnonzero(hf != ora).tolist()
            assert diff == [ts_gt[-1]] and hf[ts_gt[-1]] and not ora[ts_gt[-1]], (gen, diff)
            n_plus1 += 1
            continue
        assert np.array_equal(hf, ora), (gen, np.flatnonzero(hf != ora).tolist())
        n_same += 1
    assert n_same > 700 and n_plus1 > 25      # every class was exercised


def test_first_step_masks_and_max_initial_timestamp():
    st, o, init = _setup()
    for max_initial in (0, 1, 50, 100000):
        o2 = OW.DecodingOptions(suppress_ids=o.suppress_ids, blank_ids=o.blank_ids, max_initial_timestamp_index=max_initial)
        hf, ora = _hf_mask(st, o2, init, []), _oracle_mask(st, o2, init, [])
        assert np.array_equal(hf, ora)
        allowed = np.flatnonzero(~ora)
        assert allowed.min() >= st.timestamp_begin and allowed.max() <= min(V - 1, st.timestamp_begin + max_initial)


def test_no_timestamps_mode_only_suppresses_the_lists():
    st, o, init = _setup()
    o2 = OW.DecodingOptions(suppress_ids=o.suppress_ids, blank_ids=o.blank_ids, timestamps=False)
    base, ts = OW.rule_masks(init + [5, 6], len(init), st, o2, V)
    assert not torch.isinf(ts).any()
    assert set(np.flatnonzero(torch.isinf(base).numpy()).tolist()) == {t for t in o.suppress_ids if t < V}
