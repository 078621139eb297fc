"""Host-side mirror of CosyHiFTGenerator (TTS/CosyVoice2/HiFiGAN/CosyHiFTGenerator.swift:261-512), backed by the gfx950 HIP
layer: f0_predictor / m_source / decode / __call__ are the Swift module's public methods, same argument meaning.  The Gaussian
the reference draws inside SineGen2 is an explicit `noise` argument (None = no additive noise)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .whisper import _TensorView


class _HiftCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("base_channels", C.c_int32), ("nb_harmonics", C.c_int32), ("sampling_rate", C.c_int32),
                ("n_ups", C.c_int32), ("up_rates", C.c_int32 * 4), ("up_kernels", C.c_int32 * 4),
                ("n_res_kernels", C.c_int32), ("res_kernels", C.c_int32 * 4), ("src_res_kernels", C.c_int32 * 4),
                ("n_dilations", C.c_int32), ("dilations", C.c_int32 * 4),
                ("nsf_alpha", C.c_float), ("nsf_sigma", C.c_float), ("voiced_threshold", C.c_float), ("lrelu_slope", C.c_float),
                ("audio_limit", C.c_float)]


def _declare(lib):
    if getattr(lib, "_hift_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.mia_hift_load.restype = vp
    lib.mia_hift_load.argtypes = [vp, C.POINTER(_HiftCfg), C.POINTER(_TensorView), i32]
    lib.mia_hift_free.restype = None
    lib.mia_hift_free.argtypes = [vp]
    lib.mia_hift_upsample_factor.restype = i32
    lib.mia_hift_upsample_factor.argtypes = [vp]
    lib.mia_hift_f0.restype = i32
    lib.mia_hift_f0.argtypes = [vp, vp, i32, vp, i32]
    lib.mia_hift_source.restype = i32
    lib.mia_hift_source.argtypes = [vp, vp, i32, vp, vp, i32]
    lib.mia_hift_decode.restype = i32
    lib.mia_hift_decode.argtypes = [vp, vp, i32, vp, vp, i32]
    lib.mia_hift_vocode.restype = i32
    lib.mia_hift_vocode.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32]
    lib.mia_hift_vocode_batch.restype = i32
    lib.mia_hift_vocode_batch.argtypes = [vp, vp, vp, i32, vp, vp, i32]
    lib._hift_declared = True


def _pad4(v):
    v = list(v)
    return (C.c_int32 * 4)(*(v + [0] * (4 - len(v))))


class HiFTGenerator:
    def __init__(self, ctx, h, cfg):
        self.ctx, self.h, self.cfg = ctx, h, cfg
        ctx.adopt(self)
        self.up = ctx.lib.mia_hift_upsample_factor(h)

    @staticmethod
    def load(ctx: _lib.Context, cfg, weights: dict[str, np.ndarray]) -> "HiFTGenerator":
        _declare(ctx.lib)
        if cfg.n_fft != 16 or cfg.hop != 4:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "HiFT: only istft n_fft 16 / hop 4 is built")
        c = _HiftCfg(cfg.in_channels, cfg.base_channels, cfg.nb_harmonics, cfg.sampling_rate, len(cfg.up_rates), _pad4(cfg.up_rates),
                     _pad4(cfg.up_kernels), len(cfg.res_kernels), _pad4(cfg.res_kernels), _pad4(cfg.src_res_kernels),
                     len(cfg.dilations), _pad4(cfg.dilations), cfg.nsf_alpha, cfg.nsf_sigma, cfg.voiced_threshold, cfg.lrelu_slope,
                     cfg.audio_limit)
        views = (_TensorView * len(weights))()
        keep = []
        for i, (name, arr) in enumerate(weights.items()):
            a = np.ascontiguousarray(arr, np.float32)
            keep.append(a)
            views[i] = _TensorView(name.encode(), _lib.F32, a.ndim, (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim))), a.ctypes.data)
        h = ctx.lib.mia_hift_load(ctx.h, C.byref(c), views, len(weights))
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return HiFTGenerator(ctx, h, cfg)

    def close(self):
        if self.h and getattr(self.ctx, 'h', None):
            self.ctx.lib.mia_hift_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mel(self, mel):
        m = np.ascontiguousarray(mel, np.float32)
        if m.ndim != 2 or m.shape[0] != self.cfg.in_channels:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"HiFT: mel must be [{self.cfg.in_channels}, T]")
        return m

    def f0_predictor(self, mel: np.ndarray) -> np.ndarray:
        m = self._mel(mel)
        f0 = np.empty(m.shape[1], np.float32)
        self.ctx.check(self.ctx.lib.mia_hift_f0(self.h, m.ctypes.data, m.shape[1], f0.ctypes.data, _lib.MEM_HOST))
        return f0

    def m_source(self, f0: np.ndarray, noise: np.ndarray | None = None) -> np.ndarray:
        f = np.ascontiguousarray(f0, np.float32)
        T = f.shape[0]
        n = self._noise(noise, T)
        s = np.empty(T * self.up, np.float32)
        self.ctx.check(self.ctx.lib.mia_hift_source(self.h, f.ctypes.data, T, n.ctypes.data if n is not None else None, s.ctypes.data, _lib.MEM_HOST))
        return s

    def _noise(self, noise, T):
        if noise is None:
            return None
        n = np.ascontiguousarray(noise, np.float32)
        if n.shape != (T * self.up, self.cfg.nb_harmonics + 1):
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "HiFT: noise must be [upsample * T, nb_harmonics + 1]")
        return n

    def decode(self, mel: np.ndarray, s: np.ndarray) -> np.ndarray:
        m = self._mel(mel)
        T = m.shape[1]
        src = np.ascontiguousarray(s, np.float32)
        if src.shape != (T * self.up,):
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "HiFT: source must be [upsample * T]")
        pcm = np.empty(T * self.up, np.float32)
        self.ctx.check(self.ctx.lib.mia_hift_decode(self.h, m.ctypes.data, T, src.ctypes.data, pcm.ctypes.data, _lib.MEM_HOST))
        return pcm

    def __call__(self, mel: np.ndarray, cache_source: np.ndarray | None = None, noise: np.ndarray | None = None):
        """-> (waveform [480 T], source [480 T])   (callAsFunction / inference, :482-511)"""
        m = self._mel(mel)
        T = m.shape[1]
        n = self._noise(noise, T)
        cache = None if cache_source is None else np.ascontiguousarray(cache_source, np.float32)
        pcm = np.empty(T * self.up, np.float32)
        src = np.empty(T * self.up, np.float32)
        self.ctx.check(self.ctx.lib.mia_hift_vocode(self.h, m.ctypes.data, T, n.ctypes.data if n is not None else None,
                                                    cache.ctypes.data if cache is not None and cache.size else None,
                                                    0 if cache is None else int(cache.size), pcm.ctypes.data, src.ctypes.data, _lib.MEM_HOST))
        return pcm, src

    inference = __call__

    def vocode_batch(self, mels, noises=None):
        """mia_hift_vocode_batch: one pass over several utterances (mels: list of [80, T_u]; noises: matching list or None) ->
        list of waveforms [480 T_u], each bit-identical to self(mel_u, noise=noise_u)[0]."""
        ms = [self._mel(m) for m in mels]
        T = np.asarray([m.shape[1] for m in ms], np.int32)
        flat = np.concatenate([m.reshape(-1) for m in ms]).astype(np.float32)
        nz = None
        if noises is not None:
            nz = np.concatenate([self._noise(n, int(t)).reshape(-1) for n, t in zip(noises, T)]).astype(np.float32)
        pcm = np.empty(int(T.sum()) * self.up, np.float32)
        self.ctx.check(self.ctx.lib.mia_hift_vocode_batch(self.h, flat.ctypes.data, T.ctypes.data, len(ms), nz.ctypes.data if nz is not None else None,
                                                          pcm.ctypes.data, _lib.MEM_HOST))
        out, off = [], 0
        for t in T:
            out.append(pcm[off:off + int(t) * self.up].copy())
            off += int(t) * self.up
        return out
