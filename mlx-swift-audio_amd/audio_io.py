"""Front of the path for real files: RIFF / WAVE decoding and mixing to mono as the engines do before any tensor
work -- WhisperEngine.loadAudioFile (STT/Whisper/WhisperEngine.swift:328-369: AVAudioFile -> float32, channels AVERAGED in float32) --
and the anti-aliased rate conversion (Audio/AudioResampler.swift:15-88) on the device (mia_resample_sinc: a documented polyphase
windowed-sinc filter; Apple's AVAudioConverter is not specified in the reference's sources, so this is a stated substitute).
AVAudioFile's other containers (m4a, mp3, caf ...) need Apple's decoders and stay out of scope."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import _lib


def read_wav(path: str) -> tuple[np.ndarray, int]:
    """-> (float32 mono [n], sample_rate).  PCM 8 (unsigned) / 16 / 24 / 32 bit, IEEE float 32 / 64, WAVE_FORMAT_EXTENSIBLE; integer PCM is
    scaled by 1 / 2^(bits-1) (what Core Audio's float conversion does); channels are averaged: sum in channel order, then / count."""
    data = open(path, "rb").read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise _lib.MiaError(_lib.ERR_INVALID_AUDIO, "read_wav: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or len(fmt) < 16:
        raise _lib.MiaError(_lib.ERR_INVALID_AUDIO, "read_wav: missing fmt / data chunk")
    tag, ch, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE and len(fmt) >= 26:                     # extensible: the real format is the first two bytes of the sub-format GUID
        tag = struct.unpack("<H", fmt[24:26])[0]
    if ch < 1 or rate < 1 or align != ch * (bits // 8) or bits % 8:
        raise _lib.MiaError(_lib.ERR_INVALID_AUDIO, f"read_wav: unsupported layout (channels {ch}, bits {bits}, block align {align})")
    n = len(pcm) // align
    raw = np.frombuffer(pcm, np.uint8, n * align).reshape(n, ch, bits // 8)
    if tag == 1:
        if bits == 8:
            x = (raw[..., 0].astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = raw.reshape(n, ch * 2).view("<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            v = raw[..., 0].astype(np.int32) | (raw[..., 1].astype(np.int32) << 8) | (raw[..., 2].astype(np.int8).astype(np.int32) << 16)
            x = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (raw.reshape(n, ch * 4).view("<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise _lib.MiaError(_lib.ERR_INVALID_AUDIO, f"read_wav: {bits}-bit PCM is not supported")
    elif tag == 3 and bits in (32, 64):
        x = raw.reshape(n, ch * (bits // 8)).view("<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise _lib.MiaError(_lib.ERR_INVALID_AUDIO, f"read_wav: format tag {tag} / {bits} bit is not PCM or IEEE float")
    x = x.reshape(n, ch)
    if ch == 1:
        return np.ascontiguousarray(x[:, 0]), int(rate)
    s = np.zeros(n, np.float32)
    for c in range(ch):                                       # float32 sum in channel order, then one division (WhisperEngine.swift:355-362)
        s = s + x[:, c]
    return (s / np.float32(ch)).astype(np.float32), int(rate)


def resample(ctx: _lib.Context, audio: np.ndarray, from_rate: int, to_rate: int) -> np.ndarray:
    """AudioResampler.resample(_:from:to:) on the device; the same rate returns the input unchanged (AudioResampler.swift:20-22)."""
    a = np.ascontiguousarray(audio, np.float32).reshape(-1)
    if from_rate == to_rate:
        return a
    lib = ctx.lib
    lib.mia_resample_sinc_len.restype = C.c_int64
    lib.mia_resample_sinc_len.argtypes = [C.c_int64, C.c_int, C.c_int]
    lib.mia_resample_sinc.restype = C.c_int
    lib.mia_resample_sinc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int]
    n = int(lib.mia_resample_sinc_len(a.size, from_rate, to_rate))
    out = np.empty(max(n, 1), np.float32)
    got = C.c_int64(0)
    ctx.check(lib.mia_resample_sinc(ctx.h, a.ctypes.data, a.size, from_rate, to_rate, out.ctypes.data, out.size, C.byref(got), _lib.MEM_HOST))
    return out[:got.value]


def load_audio(ctx: _lib.Context, path: str, target_rate: int = 16000) -> np.ndarray:
    """File -> float32 mono at `target_rate`: the front half of WhisperEngine.transcribe(url:) (WhisperEngine.swift:94-143)."""
    x, rate = read_wav(path)
    return resample(ctx, x, rate, target_rate)
