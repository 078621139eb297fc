"""CPU checks of the CAM++ oracle (oracle/campplus.py): the Kaldi filterbank against an independent direct evaluation, the HTK
triangles' structure, segment pooling's zero-padded tail, and BatchNorm / shape bookkeeping of the network."""
import numpy as np
import torch

from mlx_swift_audio_amd import synthetic as S
from oracle import campplus as OC
from oracle import logmel as OL


def test_povey_window_and_filter_structure():
    w = OC.povey_window(400)
    assert w.shape == (400,) and w[0] == 0 and abs(w[200] - 1) < 1e-4 and np.allclose(w, w[::-1], atol=1e-6)
    f = OC.mel_filters_htk(16000, 512, 80, 20.0, 8000.0)
    assert f.shape == (257, 80) and f.min() >= 0 and f.max() <= 1.0
    live = f.sum(axis=0) > 0
    # rounding the edges to FFT bins (CAMPPlus.swift:151) leaves a few low triangles empty (lo == centre == hi): those bins sit at
    # log(FLT_EPSILON) for every frame, in the reference too
    assert 0 < (~live).sum() <= 6 and live[40:].all()
    assert (np.diff(f.argmax(axis=0)[live]) > 0).all()                 # centres of the live triangles strictly increase


def test_kaldi_fbank_matches_direct_evaluation():
    x = OL.synth_clip(5, 16000)[:4000]
    fb = OC.kaldi_fbank(x)
    assert fb.shape == ((4000 - 400) // 160 + 1, 80)
    # frame 3 by hand in float64: DC removal, pre-emphasis with the first sample kept, window, 512-point DFT, triangles, log
    fr = x[3 * 160:3 * 160 + 400].astype(np.float64)
    fr = fr - fr.mean()
    fr = np.concatenate([fr[:1], fr[1:] - 0.97 * fr[:-1]]) * OC.povey_window(400).astype(np.float64)
    n = np.arange(400)
    spec = np.array([np.sum(fr * np.exp(-2j * np.pi * k * n / 512)) for k in range(257)])
    want = np.log(np.maximum((np.abs(spec) ** 2) @ OC.mel_filters_htk(16000, 512, 80, 20.0, 8000.0).astype(np.float64), 1.1920929e-07))
    np.testing.assert_allclose(fb[3], want, atol=2e-4)
    assert OC.kaldi_fbank(np.zeros(400, np.float32)).shape == (1, 80)  # all-zero frame clamps to log(FLT_EPSILON)
    assert np.allclose(OC.kaldi_fbank(np.zeros(400, np.float32)), np.log(1.1920929e-07))


def test_seg_pooling_pads_before_averaging():
    x = torch.arange(250, dtype=torch.float32).reshape(1, 1, 250)
    s = OC.CAMPPlusOracle.seg_pooling(x)
    assert s.shape == (1, 1, 250)
    assert abs(float(s[0, 0, 0]) - 49.5) < 1e-4 and abs(float(s[0, 0, 150]) - 149.5) < 1e-4
    assert abs(float(s[0, 0, 249]) - float(x[0, 0, 200:].sum()) / 100.0) < 1e-3      # 50 live frames still divide by 100


def test_network_shapes_and_determinism():
    w = S.campplus_weights(1)
    assert w["tdnn.linear.weight"].shape == (128, 5, 320) and w["dense.linear.weight"].shape == (192, 1, 1024)
    assert w["blocks.1.layers.23.linear1.weight"].shape == (128, 1, 256 + 23 * 32)
    assert "dense.nonlinear.0.weight" not in w                         # BatchNorm(affine: false)
    o = OC.CAMPPlusOracle(w)
    x = OL.synth_clip(2, 16000)
    e1, e2 = o.inference(x), o.inference(x)
    assert e1.shape == (1, 192) and np.isfinite(e1).all() and np.array_equal(e1, e2)
    e3 = o.inference(OL.synth_clip(3, 16000))
    assert np.abs(e1 - e3).max() > 1e-3                                # the embedding depends on the clip
