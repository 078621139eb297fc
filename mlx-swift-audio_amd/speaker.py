"""Host-side mirror of CAMPlusSpeakerEncoder (TTS/CosyVoice2/SpeakerEncoder/CAMPlusSpeakerEncoder.swift:12-150) over the gfx950
HIP layer: `__call__(audio16k) -> [1, 192]` (zeros when no weights are loaded, as the reference), `extract_fbank`, and the
underlying CAMPPlus forward on precomputed features (Codec/S3Gen/CAMPPlus.swift:755-785)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .whisper import _TensorView

EMBEDDING_DIM = 192


def _declare(lib):
    if getattr(lib, "_campplus_declared", False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.mia_campplus_load.restype = vp
    lib.mia_campplus_load.argtypes = [vp, C.POINTER(_TensorView), i32]
    lib.mia_campplus_free.restype = None
    lib.mia_campplus_free.argtypes = [vp]
    lib.mia_kaldi_fbank_frames.restype = i64
    lib.mia_kaldi_fbank_frames.argtypes = [i64]
    lib.mia_campplus_fbank.restype = i32
    lib.mia_campplus_fbank.argtypes = [vp, vp, i64, i32, vp, i32]
    lib.mia_campplus_forward.restype = i32
    lib.mia_campplus_forward.argtypes = [vp, vp, i32, vp, i32]
    lib.mia_campplus_embed.restype = i32
    lib.mia_campplus_embed.argtypes = [vp, vp, i64, vp, i32]
    lib._campplus_declared = True


class CAMPlusSpeakerEncoder:
    embedding_dim = EMBEDDING_DIM

    def __init__(self, ctx, h=None):
        self.ctx, self.h = ctx, h
        if h:
            ctx.adopt(self)

    @property
    def is_loaded(self) -> bool:
        return bool(self.h)

    @staticmethod
    def load(ctx: _lib.Context, weights: dict[str, np.ndarray]) -> "CAMPlusSpeakerEncoder":
        """weights: the checkpoint's campplus tensors; a "campplus." key prefix is stripped (sanitizeWeights, :93-104).  An empty
        dict gives the reference's unloaded encoder (zero embeddings, :109-116)."""
        _declare(ctx.lib)
        weights = {(k[len("campplus."):] if k.startswith("campplus.") else k): v for k, v in weights.items()}
        if not weights:
            return CAMPlusSpeakerEncoder(ctx, None)
        views = (_TensorView * len(weights))()
        keep = []
        for i, (name, arr) in enumerate(weights.items()):
            a = np.ascontiguousarray(arr, np.float32)
            keep.append(a)
            views[i] = _TensorView(name.encode(), _lib.F32, a.ndim, (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim))), a.ctypes.data)
        h = ctx.lib.mia_campplus_load(ctx.h, views, len(weights))
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return CAMPlusSpeakerEncoder(ctx, h)

    def close(self):
        if self.h and getattr(self.ctx, "h", None):
            self.ctx.lib.mia_campplus_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _need(self):
        if not self.h:
            raise _lib.MiaError(_lib.ERR_MODEL_NOT_LOADED, "CAM++ weights are not loaded")

    def extract_fbank(self, audio, mean_norm: bool = False) -> np.ndarray:
        """kaldiFbankCAMPPlus: 16 kHz mono [n] -> [frames, 80]."""
        self._need()
        x = np.ascontiguousarray(np.asarray(audio, np.float32).reshape(-1))
        F = self.ctx.lib.mia_kaldi_fbank_frames(x.size)
        out = np.empty((max(F, 0), 80), np.float32)
        self.ctx.check(self.ctx.lib.mia_campplus_fbank(self.h, x.ctypes.data, x.size, int(mean_norm), out.ctypes.data, _lib.MEM_HOST))
        return out

    def forward(self, feats) -> np.ndarray:
        """CAMPPlus.callAsFunction on one clip's features [T, 80] -> [192]."""
        self._need()
        f = np.ascontiguousarray(feats, np.float32)
        if f.ndim != 2 or f.shape[1] != 80:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "CAM++ features must be [frames, 80]")
        out = np.empty(EMBEDDING_DIM, np.float32)
        self.ctx.check(self.ctx.lib.mia_campplus_forward(self.h, f.ctypes.data, f.shape[0], out.ctypes.data, _lib.MEM_HOST))
        return out

    def __call__(self, audio, sample_rate: int = 16000) -> np.ndarray:
        """16 kHz mono clip [n] -> [1, 192]; zeros when no weights are loaded (CAMPlusSpeakerEncoder.swift:109-116)."""
        if not self.h:
            return np.zeros((1, EMBEDDING_DIM), np.float32)
        x = np.ascontiguousarray(np.asarray(audio, np.float32).reshape(-1))
        out = np.empty((1, EMBEDDING_DIM), np.float32)
        self.ctx.check(self.ctx.lib.mia_campplus_embed(self.h, x.ctypes.data, x.size, out.ctypes.data, _lib.MEM_HOST))
        return out
