"""WAV decoding / mono mixing (CPU) and the device resampler (GPU) of audio_io.py.  The resampler is a documented substitute for
AVAudioConverter (no parity claim): it is checked against its own published definition evaluated in float64 on the host, and through
signal properties (unit DC gain, pass-band tone preserved, stop-band tone rejected, output length)."""
import ctypes as C
import struct

import numpy as np
import pytest


def _wav(path, samples, rate, bits, fmt_tag=1, extensible=False):
    """samples float [n, ch] in [-1, 1) -> a RIFF file written by hand (no library), little-endian."""
    n, ch = samples.shape
    if fmt_tag == 3:
        body = samples.astype("<f4" if bits == 32 else "<f8").tobytes()
    elif bits == 8:
        body = np.clip(np.round(samples * 128.0) + 128, 0, 255).astype(np.uint8).tobytes()
    elif bits == 16:
        body = np.clip(np.round(samples * 32768.0), -32768, 32767).astype("<i2").tobytes()
    elif bits == 24:
        v = np.clip(np.round(samples * 8388608.0), -8388608, 8388607).astype(np.int32).reshape(-1)
        body = b"".join(int(x).to_bytes(3, "little", signed=True) for x in v)
    else:
        body = np.clip(np.round(samples.astype(np.float64) * 2147483648.0), -2147483648, 2147483647).astype("<i4").tobytes()
    align = ch * bits // 8
    if extensible:
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * align, align, bits, 22, bits, 0) + struct.pack("<H", fmt_tag) + b"\x00" * 14
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, rate, rate * align, align, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + b"data" + struct.pack("<I", len(body)) + body
    open(path, "wb").write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


@pytest.mark.parametrize("bits,tag,ext", [(8, 1, False), (16, 1, False), (24, 1, False), (32, 1, False), (32, 3, False), (64, 3, False), (16, 1, True)])
def test_read_wav_formats_and_mono_mix(tmp_path, bits, tag, ext):
    from mlx_swift_audio_amd import audio_io as IO
    rng = np.random.default_rng(bits)
    x = (0.8 * rng.uniform(-1, 1, (301, 3))).astype(np.float32)
    p = str(tmp_path / "a.wav")
    _wav(p, x, 22050, bits, tag, ext)
    got, rate = IO.read_wav(p)
    assert rate == 22050 and got.dtype == np.float32 and got.shape == (301,)
    step = {8: 1 / 128, 16: 1 / 32768, 24: 1 / 8388608, 32: 1e-7, 64: 1e-7}[bits]
    want = ((x[:, 0] + x[:, 1]) + x[:, 2]) / np.float32(3)
    assert np.abs(got - want).max() <= step + 1e-6
    _wav(p, x[:, :1], 8000, bits, tag, ext)                        # mono: no mixing
    mono, rate = IO.read_wav(p)
    assert rate == 8000 and np.abs(mono - x[:, 0]).max() <= step / 2 + 1e-6


def test_read_wav_rejects_garbage(tmp_path):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import audio_io as IO
    p = str(tmp_path / "b.wav")
    open(p, "wb").write(b"RIFF\x04\x00\x00\x00WAVX")
    with pytest.raises(m.MiaError):
        IO.read_wav(p)
    _wav(p, np.zeros((4, 1), np.float32), 8000, 16, fmt_tag=7)        # mu-law tag: not PCM
    with pytest.raises(m.MiaError):
        IO.read_wav(p)


def _table(lib, fr, to):
    lib.mia_resample_sinc_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L, taps = C.c_int(0), C.c_int(0)
    assert lib.mia_resample_sinc_table(fr, to, None, 0, C.byref(L), C.byref(taps)) == 0
    t = np.zeros((L.value, taps.value), np.float32)
    assert lib.mia_resample_sinc_table(fr, to, t.ctypes.data, t.size, None, None) == 0
    return t


def test_sinc_table_is_the_published_filter():
    """Host arithmetic (runs without a GPU): every phase has unit DC gain, the table is the Kaiser-windowed sinc of resample.hip's header, and
    the low-pass sits at 0.945 x the lower Nyquist."""
    import mlx_swift_audio_amd as m
    lib = m._lib.load()
    for fr, to in ((48000, 16000), (44100, 16000), (16000, 24000), (22050, 24000)):
        t = _table(lib, fr, to)
        g = np.gcd(fr, to)
        assert t.shape[0] == to // g and t.shape[1] % 2 == 1
        np.testing.assert_allclose(t.sum(1), 1.0, atol=2e-6)
        fc = 0.5 * 0.945 * min(1.0, to / fr)
        hwd = 16.0 / (2 * fc)
        hw = (t.shape[1] - 1) // 2
        p = t.shape[0] // 3
        d = np.arange(-hw, hw + 1) - p / t.shape[0]
        r = d / hwd
        w = np.where(np.abs(r) < 1, 2 * fc * np.sinc(2 * fc * d) * np.i0(8.6 * np.sqrt(np.clip(1 - r * r, 0, 1))) / np.i0(8.6), 0.0)
        np.testing.assert_allclose(t[p], w / w.sum(), atol=1e-6)
    up = _table(lib, 16000, 48000)
    hw = (up.shape[1] - 1) // 2
    assert abs(up[0, hw] - 0.945) < 2e-3          # the cut-off sits BELOW Nyquist (roll-off 0.945): on-grid outputs are low-passed too, by design


@pytest.mark.gpu
@pytest.mark.parametrize("fr,to", [(48000, 16000), (44100, 16000), (16000, 24000), (24000, 16000)])
def test_resample_device_matches_definition_and_rejects_aliases(ctx, fr, to):
    from mlx_swift_audio_amd import audio_io as IO
    n = 20000 + 37
    t = np.arange(n) / fr
    nyq = min(fr, to) / 2
    lo, hi = 0.4 * nyq, min(1.45 * nyq, 0.49 * fr)
    x = (0.5 * np.sin(2 * np.pi * lo * t) + 0.25).astype(np.float32)
    y = IO.resample(ctx, x, fr, to)
    g = np.gcd(fr, to)
    assert y.shape == (n * (to // g) // (fr // g),)
    # definition in float64 on the host from the same phase table
    tab = _table(ctx.lib, fr, to).astype(np.float64)
    L, M, hw = to // g, fr // g, (tab.shape[1] - 1) // 2
    xp = np.concatenate([np.zeros(hw + 2), x.astype(np.float64), np.zeros(hw + 2)])
    for j in (0, 1, 7, len(y) // 2, len(y) - 2, len(y) - 1):
        i0, p = (j * M) // L, (j * M) % L
        want = float(np.dot(tab[p], xp[i0 + 2:i0 + 2 + 2 * hw + 1]))
        assert abs(y[j] - want) < 2e-6, (j, y[j], want)
    mid = y[200:-200]
    tt = (np.arange(len(y)) / to)[200:-200]
    np.testing.assert_allclose(mid, 0.5 * np.sin(2 * np.pi * lo * tt) + 0.25, atol=2e-3)     # pass band + DC preserved
    if fr > to:                                                                              # a tone above the new Nyquist must not alias in
        z = IO.resample(ctx, np.sin(2 * np.pi * hi * t).astype(np.float32), fr, to)[200:-200]
        assert np.sqrt(np.mean(z ** 2)) < 10 ** (-60 / 20)
    assert np.array_equal(IO.resample(ctx, x, fr, fr), x)
