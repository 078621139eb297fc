"""CPU tests pinning the log-mel oracle (oracle/logmel.py) against independent float64 math.

The reference holds no golden vectors for this path: "parity unpinned" at tensor level.
These tests check the restatement against a direct float64 DFT and against properties the Swift source
states (frame counts, window symmetry, Slaney normalisation).
"""
import numpy as np
import pytest

from oracle import logmel as O


def _direct_logmel_f64(audio, n_mels, padding, window):
    a = np.concatenate([audio.astype(np.float64), np.zeros(padding)])
    a = np.concatenate([a[1:201][::-1], a, a[-201:-1][::-1]])
    nfr = 1 + (len(a) - 400) // 160
    k = np.arange(201)[:, None] * np.arange(400)[None, :]
    E = np.exp(-2j * np.pi * (k % 400) / 400.0)
    out = np.empty((nfr - 1, n_mels))
    fb = O.mel_filters(16000, 400, n_mels, 0.0, 8000.0).astype(np.float64)
    for f in range(nfr - 1):
        fr = a[f * 160:f * 160 + 400] * window.astype(np.float64)
        X = E @ fr
        out[f] = fb @ (np.abs(X) ** 2)
    ls = np.log10(np.maximum(out, 1e-10))
    ls = np.maximum(ls, ls.max() - 8.0)
    return (ls + 4.0) / 4.0


def test_windows():
    w = O.whisper_hann_window(400)
    assert w.dtype == np.float32 and w.shape == (400,)
    assert abs(w[0]) < 1e-7 and abs(w[-1]) < 1e-6          # symmetric: both ends ~0
    np.testing.assert_allclose(w, w[::-1], atol=1e-6)
    p = O.periodic_hann_window(400)
    assert abs(p[0]) < 1e-7 and p[-1] > 1e-5                # periodic: last sample not 0
    np.testing.assert_allclose(p, 0.5 * (1 - np.cos(2 * np.pi * np.arange(400) / 400)), atol=2e-6)


def test_reflect_pad_matches_numpy():
    x = np.arange(1000, dtype=np.float32)
    np.testing.assert_array_equal(O.reflect_pad(x, 200), np.pad(x, 200, mode="reflect"))
    short = np.arange(50, dtype=np.float32)                 # shorter than the padding: looped branch
    r = O.reflect_pad(short, 200)
    assert r.shape == (450,) and np.array_equal(r[200:250], short)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filters(n_mels):
    fb = O.mel_filters(16000, 400, n_mels, 0.0, 8000.0)
    assert fb.shape == (n_mels, 201) and fb.dtype == np.float32
    assert (fb >= 0).all() and fb[:, 0].sum() == 0
    # every filter is a single contiguous triangle
    for m in range(n_mels):
        nz = np.nonzero(fb[m])[0]
        assert len(nz) >= 1 and np.array_equal(nz, np.arange(nz[0], nz[-1] + 1))
    # agrees with the transformers (HF) slaney filterbank, an independent implementation in this image
    from transformers.audio_utils import mel_filter_bank
    hf = mel_filter_bank(201, n_mels, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
    np.testing.assert_allclose(fb, hf, atol=2e-5)


@pytest.mark.parametrize("n_mels,padding", [(80, 0), (128, 1600)])
def test_whisper_logmel_vs_direct_f64(n_mels, padding):
    rng = np.random.default_rng(0)
    t = np.arange(16000) / 16000.0
    audio = (0.1 * rng.standard_normal(16000) + 0.3 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    got = O.whisper_log_mel_spectrogram(audio, n_mels, padding)
    assert got.shape == ((16000 + padding) // 160, n_mels)
    ref = _direct_logmel_f64(audio, n_mels, padding, O.whisper_hann_window(400))
    np.testing.assert_allclose(got, ref, atol=2e-4)


def test_frame_counts_match_reference_constants():
    # WhisperAudio.swift:18-20: 480000 samples -> 3000 frames; as called (WhisperSTT.swift:140) -> 6000
    x = np.zeros(480000, np.float32)
    x[1000] = 1.0
    assert O.whisper_log_mel_spectrogram(x, 80).shape == (3000, 80)
    m = O.whisper_log_mel_spectrogram(x, 128, padding=480000)
    assert m.shape == (6000, 128)
    # silence frames sit at the clamp floor: (max-8+4)/4
    assert np.allclose(m[5000], m[5999])


def test_s3_variant_layout():
    x = O.synth_clip(0, 16000)
    m = O.s3_log_mel_spectrogram(x, 128)
    assert m.shape == (128, 100)
