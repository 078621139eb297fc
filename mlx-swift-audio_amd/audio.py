"""Host-side mirror of the reference's DSP front-end interface, backed by the gfx950 HIP layer.

Function names and argument meaning follow STT/Whisper/WhisperAudio.swift (whisperLogMelSpectrogram,
padOrTrim) and Codec/S3Tokenizer/S3TokenizerUtils.swift (logMelSpectrogram) so the parity tests read like
the reference's own.  All arithmetic runs in lib/libmia.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE      # WhisperAudio.swift:19
N_FRAMES = N_SAMPLES // HOP_LENGTH          # WhisperAudio.swift:20

_NP_DT = {_lib.F32: np.float32, _lib.F16: np.float16, _lib.BF16: np.uint16}


def pad_or_trim(array: np.ndarray, length: int = N_SAMPLES) -> np.ndarray:
    """padOrTrim (WhisperAudio.swift:54-67): pure host-side reshaping, no arithmetic."""
    n = array.shape[0]
    if n > length:
        return array[:length]
    if n < length:
        return np.concatenate([array, np.zeros(length - n, array.dtype)])
    return array


def _as_batch(audio):
    if isinstance(audio, np.ndarray) and audio.ndim == 1:
        clips, single = [audio], True
    else:
        clips, single = list(audio), False
    if len(clips) == 0:
        raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "empty batch")
    clips = [np.ascontiguousarray(c, np.float32) for c in clips]
    offs = np.zeros(len(clips) + 1, np.int64)
    np.cumsum([c.shape[0] for c in clips], out=offs[1:])
    return clips, offs, single


def _run(fn_name, ctx, audio, n_mels, padding, n_frames, dtype, channel_major):
    clips, offs, single = _as_batch(audio)
    B = len(clips)
    pcm = np.concatenate(clips) if B > 1 else clips[0]
    if n_frames is None:
        lens = {(int(offs[b + 1] - offs[b]) + padding) // HOP_LENGTH for b in range(B)}
        if len(lens) != 1:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "ragged batch needs an explicit n_frames")
        n_frames = lens.pop()
    if n_frames <= 0:
        raise _lib.MiaError(_lib.ERR_INVALID_AUDIO, "Input is too short for STFT")
    shape = (B, n_mels, n_frames) if channel_major else (B, n_frames, n_mels)
    out = np.empty(shape, _NP_DT[dtype])
    fn = getattr(ctx.lib, fn_name)
    ctx.check(fn(ctx.h, pcm.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), B, n_mels, padding,
                 n_frames, out.ctypes.data_as(C.c_void_p), dtype, _lib.MEM_HOST))
    return out[0] if single else out


def whisper_log_mel_spectrogram(ctx: _lib.Context, audio, n_mels: int, padding: int = 0, n_frames: int | None = None,
                                dtype: int = _lib.F32):
    """whisperLogMelSpectrogram(audio:nMels:padding:) (WhisperAudio.swift:78-137).

    audio: float32 [T] or a list of clips.  Returns [n_frames, n_mels] (or [B, n_frames, n_mels]); with
    n_frames=None every frame of the padded utterance is returned, exactly like the reference."""
    return _run("mia_logmel_whisper", ctx, audio, n_mels, padding, n_frames, dtype, False)


def s3_log_mel_spectrogram(ctx: _lib.Context, audio, n_mels: int = 128, padding: int = 0, n_frames: int | None = None,
                           dtype: int = _lib.F32):
    """logMelSpectrogram / logMelSpectrogramChatterbox (S3TokenizerUtils.swift:102-208): [n_mels, frames]."""
    return _run("mia_logmel_s3", ctx, audio, n_mels, padding, n_frames, dtype, True)


def bf16_to_f32(a: np.ndarray) -> np.ndarray:
    return (a.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bf16 bit pattern (uint16)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) >> 16).astype(np.uint16)


def s3gen_mel_spectrogram(ctx: _lib.Context, y) -> np.ndarray:
    """s3genMelSpectrogram(y:) with its defaults (Codec/S3Gen/Mel/S3GenMel.swift:43-102): 24 kHz mono float32 [T] -> [80, frames]."""
    import ctypes as C
    lib = ctx.lib
    if not getattr(lib, "_mel24_declared", False):
        lib.mia_mel_s3gen_frames.restype = C.c_int64
        lib.mia_mel_s3gen_frames.argtypes = [C.c_int64]
        lib.mia_mel_s3gen.restype = C.c_int
        lib.mia_mel_s3gen.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        lib._mel24_declared = True
    a = np.ascontiguousarray(y, np.float32).reshape(-1)
    frames = max(int(lib.mia_mel_s3gen_frames(a.shape[0])), 0)
    out = np.empty((80, frames), np.float32)
    ctx.check(lib.mia_mel_s3gen(ctx.h, a.ctypes.data, a.shape[0], out.ctypes.data, _lib.MEM_HOST))
    return out


def resample_audio(ctx: _lib.Context, audio, from_rate: int, to_rate: int) -> np.ndarray:
    """resampleAudio (TTS/CosyVoice2/CosyVoice2TTS.swift:733-744): plain linear interpolation, ratio in float32."""
    import ctypes as C
    a = np.ascontiguousarray(audio, np.float32).reshape(-1)
    if from_rate == to_rate:
        return a
    lib = ctx.lib
    if not getattr(lib, "_rs_declared", False):
        lib.mia_resample_linear_len.restype = C.c_int64
        lib.mia_resample_linear_len.argtypes = [C.c_int64, C.c_float]
        lib.mia_resample_linear.restype = C.c_int
        lib.mia_resample_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int]
        lib._rs_declared = True
    ratio = float(np.float32(to_rate) / np.float32(from_rate))
    out = np.empty(int(lib.mia_resample_linear_len(a.shape[0], ratio)), np.float32)
    ctx.check(lib.mia_resample_linear(ctx.h, a.ctypes.data, a.shape[0], ratio, out.ctypes.data, _lib.MEM_HOST))
    return out
