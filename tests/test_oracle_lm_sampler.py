"""Independent pin of the oracle's Orpheus sampler front half (row a12): repetition penalty -> temperature -> top-p
(OrpheusTTS.swift:388-461, restated in oracle/lm.py:top_p_filter) against `transformers`' RepetitionPenaltyLogitsProcessor,
TemperatureLogitsWarper and TopPLogitsWarper.  Same published definitions; the only difference is the nucleus edge at an EXACT tie
(the port keeps a token whose preceding cumulative mass equals top_p, transformers drops it), which random float32 logits do not hit."""
import numpy as np
import pytest
import torch

from oracle import lm as OL

transformers = pytest.importorskip("transformers")
from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopPLogitsWarper  # noqa: E402


@pytest.mark.parametrize("seed", range(6))
def test_top_p_filter_matches_transformers(seed):
    rng = np.random.default_rng(seed)
    V = 700
    for _ in range(40):
        logits = (rng.standard_normal(V) * rng.uniform(0.5, 6.0)).astype(np.float32)
        history = rng.integers(0, V, int(rng.integers(0, 24))).tolist()
        rep = float(rng.choice([1.0, 1.1, 1.3, 2.0]))
        temp = float(rng.choice([0.3, 0.6, 1.0, 1.7]))
        top_p = float(rng.choice([0.5, 0.8, 0.9, 0.97]))
        got = OL.top_p_filter(logits, history, rep, temp, top_p)
        scores = torch.from_numpy(logits)[None].clone()
        ids = torch.tensor([history], dtype=torch.long)
        if rep != 1.0 and history:
            scores = RepetitionPenaltyLogitsProcessor(rep)(ids, scores)
        scores = TemperatureLogitsWarper(temp)(ids, scores)
        scores = TopPLogitsWarper(top_p, min_tokens_to_keep=1)(ids, scores)
        want = scores[0].numpy()
        assert np.array_equal(np.isfinite(got), np.isfinite(want)), (rep, temp, top_p)
        keep = np.isfinite(want)
        np.testing.assert_allclose(got[keep], want[keep], rtol=1e-6, atol=1e-6)
        # the nucleus is the smallest prefix (by probability) whose mass exceeds top_p
        p = np.exp(want[keep] - want[keep].max()); full = np.exp((logits_scaled := _scaled(logits, history, rep, temp)) - logits_scaled.max())
        mass = p.sum() / full.sum()
        assert mass >= top_p - 1e-6


def _scaled(logits, history, rep, temp):
    lg = logits.astype(np.float64).copy()
    for t in set(history):
        lg[t] = lg[t] * rep if lg[t] < 0 else lg[t] / rep
    return lg / temp
