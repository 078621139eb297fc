"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference log-mel front end.

parity unpinned: the reference holds no tensor-level golden vectors for this path;
this restatement follows the Swift source line by line and is cross-checked against an independent
float64 DFT in tests/test_oracle_logmel.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to /root/reference/package):
  STT/Whisper/WhisperAudio.swift:32-44    whisperHannWindow   (symmetric Hann)
  STT/Whisper/WhisperAudio.swift:54-67    padOrTrim
  STT/Whisper/WhisperAudio.swift:78-137   whisperLogMelSpectrogram
  Codec/S3Tokenizer/S3TokenizerUtils.swift:213-221  hanningWindow
  Codec/S3Tokenizer/S3TokenizerUtils.swift:224-263  stft
  Codec/S3Tokenizer/S3TokenizerUtils.swift:266-298  reflectPad
  Codec/S3Tokenizer/S3TokenizerUtils.swift:301-375  melFilters
  Codec/S3Tokenizer/S3TokenizerUtils.swift:102-149,160-208  logMelSpectrogram / logMelSpectrogramChatterbox
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_SAMPLES = 480000
N_FRAMES = 3000

f32 = np.float32


def whisper_hann_window(length: int = N_FFT) -> np.ndarray:
    """WhisperAudio.swift:32-44 -- 0.5*(1-cos(2*pi*n/(N-1))) evaluated in float32."""
    if length == 1:
        return np.ones(1, f32)
    n = np.arange(length, dtype=f32)
    factor = f32(2.0) * f32(np.pi) / f32(length - 1)
    return (f32(0.5) * (f32(1.0) - np.cos(n * factor, dtype=f32))).astype(f32)


def hanning_window(length: int) -> np.ndarray:
    """S3TokenizerUtils.swift:213-221 -- symmetric Hann written as 0.5+0.5*cos(pi*(1-L+2i)/(L-1))."""
    if length == 1:
        return np.ones(1, f32)
    n = np.arange(1 - length, length, 2, dtype=f32)
    factor = f32(np.pi) / f32(length - 1)
    return (f32(0.5) + f32(0.5) * np.cos(n * factor, dtype=f32)).astype(f32)


def periodic_hann_window(length: int) -> np.ndarray:
    """S3TokenizerUtils.swift:117,172 -- hanningWindow(N+1)[0..<N]."""
    return hanning_window(length + 1)[:length].copy()


def reflect_pad(x: np.ndarray, padding: int) -> np.ndarray:
    """S3TokenizerUtils.swift:266-298 (no edge repeat)."""
    if padding == 0:
        return x
    n = x.shape[0]
    if n == 1:
        return np.concatenate([np.full(padding, x[0], x.dtype), x, np.full(padding, x[0], x.dtype)])
    prefix = x[1:min(padding + 1, n)][::-1]
    suffix = x[max(0, n - padding - 1):n - 1][::-1]
    while prefix.shape[0] < padding:
        additional = min(padding - prefix.shape[0], n - 1)
        prefix = np.concatenate([x[1:additional + 1][::-1], prefix])
    while suffix.shape[0] < padding:
        additional = min(padding - suffix.shape[0], n - 1)
        suffix = np.concatenate([suffix, x[n - additional - 1:n - 1][::-1]])
    return np.concatenate([prefix[:padding], x, suffix[:padding]])


def stft(x: np.ndarray, window: np.ndarray, n_fft: int, hop: int, center: bool = True) -> np.ndarray:
    """S3TokenizerUtils.swift:224-263 -- frames via strided view, window multiply, unnormalised rfft.
    Returns complex64 [num_frames, n_fft//2+1]."""
    x = np.asarray(x, f32)
    w = np.asarray(window, f32)
    if w.shape[0] < n_fft:
        w = np.concatenate([w, np.zeros(n_fft - w.shape[0], f32)])
    if center:
        x = reflect_pad(x, n_fft // 2)
    num_frames = 1 + (x.shape[0] - n_fft) // hop
    if num_frames <= 0:
        raise ValueError("Input is too short for STFT")
    frames = np.lib.stride_tricks.as_strided(x, (num_frames, n_fft), (hop * x.itemsize, x.itemsize))
    windowed = (frames * w).astype(f32)
    # MLX rfft on float32 input computes in single precision; numpy promotes to double. Round back.
    return np.fft.rfft(windowed, axis=-1).astype(np.complex64)


def mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float = 0.0, f_max: float | None = None) -> np.ndarray:
    """S3TokenizerUtils.swift:301-375 -- Slaney scale + Slaney norm, all scalar float32 host math.
    Returns float32 [n_mels, n_fft//2+1]."""
    actual_fmax = f32(f_max) if f_max is not None else f32(sample_rate) / f32(2.0)
    f_sp = f32(200.0) / f32(3.0)
    min_log_hz = f32(1000.0)
    min_log_mel = min_log_hz / f_sp
    logstep = f32(np.log(f32(6.4))) / f32(27.0)

    def hz_to_mel(hz):
        hz = f32(hz)
        if hz >= min_log_hz:
            return f32(min_log_mel + f32(np.log(f32(hz / min_log_hz))) / logstep)
        return f32(hz / f_sp)

    def mel_to_hz(mel):
        mel = f32(mel)
        if mel >= min_log_mel:
            return f32(min_log_hz * f32(np.exp(f32(logstep * f32(mel - min_log_mel)))))
        return f32(f_sp * mel)

    mel_min = hz_to_mel(f_min)
    mel_max = hz_to_mel(actual_fmax)
    pts = [mel_to_hz(f32(mel_min + f32(f32(i) * f32(mel_max - mel_min)) / f32(n_mels + 1))) for i in range(n_mels + 2)]
    nb = n_fft // 2 + 1
    freqs = [f32(f32(i) * f32(sample_rate)) / f32(n_fft) for i in range(nb)]
    fb = np.zeros((n_mels, nb), f32)
    for m in range(n_mels):
        f_left, f_center, f_right = pts[m], pts[m + 1], pts[m + 2]
        for k in range(nb):
            fr = freqs[k]
            if f_left <= fr <= f_center:
                fb[m, k] = f32(fr - f_left) / f32(f_center - f_left)
            elif f_center < fr <= f_right:
                fb[m, k] = f32(f_right - fr) / f32(f_right - f_center)
        enorm = f32(2.0) / f32(pts[m + 2] - pts[m])
        fb[m, :] = (fb[m, :] * enorm).astype(f32)
    return fb


def pad_or_trim(x: np.ndarray, length: int = N_SAMPLES) -> np.ndarray:
    """WhisperAudio.swift:54-67."""
    n = x.shape[0]
    if n > length:
        return x[:length]
    if n < length:
        return np.concatenate([x, np.zeros(length - n, x.dtype)])
    return x


def whisper_log_mel_spectrogram(audio: np.ndarray, n_mels: int, padding: int = 0) -> np.ndarray:
    """WhisperAudio.swift:78-137. audio float32 [T] -> float32 [T'/160, n_mels] (time-major)."""
    a = np.asarray(audio, f32)
    if padding > 0:
        a = np.concatenate([a, np.zeros(padding, f32)])
    window = whisper_hann_window(N_FFT)
    spec = stft(a, window, N_FFT, HOP)
    freqs = spec[:-1, :]                                   # drop the last TIME frame (:105)
    mags = (np.abs(freqs).astype(f32) ** 2).astype(f32)    # pow(abs(X), 2) (:109)
    filters = mel_filters(SAMPLE_RATE, N_FFT, n_mels, 0.0, 8000.0)
    mel = (mags @ filters.T).astype(f32)                   # (:127)
    log_spec = np.log10(np.maximum(mel, f32(1e-10))).astype(f32)
    log_spec = np.maximum(log_spec, log_spec.max() - f32(8.0))
    return ((log_spec + f32(4.0)) / f32(4.0)).astype(f32)


def s3_log_mel_spectrogram(audio: np.ndarray, n_mels: int = 128, padding: int = 0) -> np.ndarray:
    """S3TokenizerUtils.swift:102-149 / 160-208 (both variants share this arithmetic): periodic Hann,
    drop last frame, power, mel, log10 clamp, max-8, (x+4)/4.  Returns float32 [n_mels, frames]."""
    a = np.asarray(audio, f32)
    if padding > 0:
        a = np.concatenate([a, np.zeros(padding, f32)])
    window = periodic_hann_window(N_FFT)
    spec = stft(a, window, N_FFT, HOP)
    mags = (np.abs(spec[:-1, :]).astype(f32) ** 2).astype(f32)
    filters = mel_filters(SAMPLE_RATE, N_FFT, n_mels)
    mel = (mags @ filters.T).astype(f32).T
    log_spec = np.log10(np.maximum(mel, f32(1e-10))).astype(f32)
    log_spec = np.maximum(log_spec, log_spec.max() - f32(8.0))
    return ((log_spec + f32(4.0)) / f32(4.0)).astype(f32)


def synth_clip(i: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """SURVEY.md section 8d synthetic clip i (generator shared with the benchmark)."""
    from mlx_swift_audio_amd.synthetic import synth_clip as _sc
    return _sc(i, n_samples)


def s3gen_mel_spectrogram(y: np.ndarray, n_fft: int = 1920, num_mels: int = 80, sampling_rate: int = 24000, hop_size: int = 480,
                          win_size: int = 1920, fmin: float = 0.0, fmax: float = 8000.0) -> np.ndarray:
    """Codec/S3Gen/Mel/S3GenMel.swift:43-102 -- reflect pad (n_fft - hop)/2, periodic Hann, |rfft| (center false), slaney
    filterbank, log(max(., 1e-5)).  y [T] -> [num_mels, frames]."""
    y = reflect_pad(np.asarray(y, f32), (n_fft - hop_size) // 2)
    window = hanning_window(win_size + 1)[:win_size]
    spec = stft(y, window, n_fft, hop_size, center=False)
    mag = np.abs(spec).astype(f32)
    filters = mel_filters(sampling_rate, n_fft, num_mels, fmin, fmax)
    mel = (mag @ filters.T).astype(f32).T
    return np.log(np.maximum(mel, f32(1e-5))).astype(f32)
