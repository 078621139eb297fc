"""Independent pin of the word-timestamp primitives the oracle restates from WhisperTiming.swift (row f3): dynamic time warping with
its tie rule (:46-121) and the width-7 reflect-padded median filter (:227-247), against `transformers`' own ports of the published
algorithm (`generation_whisper._dynamic_time_warping`, `_median_filter`)."""
import numpy as np
import pytest
import torch

from oracle import whisper as OW

pytest.importorskip("transformers")
from transformers.models.whisper import generation_whisper as G  # noqa: E402


@pytest.mark.parametrize("seed", range(4))
def test_dtw_matches_transformers(seed):
    rng = np.random.default_rng(seed)
    for _ in range(12):
        n, m = int(rng.integers(1, 14)), int(rng.integers(1, 60))
        cost = rng.standard_normal((n, m)).astype(np.float32)
        if rng.random() < 0.5:
            cost = np.round(cost * 2) / 2                      # plenty of exact ties: the tie rule decides the path
        ti, fi = OW.dtw(cost)
        hti, hfi = G._dynamic_time_warping(cost)
        assert list(ti) == hti.tolist() and list(fi) == hfi.tolist()


def test_median_filter7_matches_transformers():
    rng = np.random.default_rng(0)
    for F in (4, 7, 8, 33, 150):
        w = rng.standard_normal((2, 3, 5, F)).astype(np.float32)
        want = G._median_filter(torch.from_numpy(w), 7).numpy()
        np.testing.assert_array_equal(OW.median_filter7(w), want)
