"""Host-side mirror of S3TokenizerV2 / V3 (Codec/S3Tokenizer/S3Tokenizer.swift:442-658), backed by the gfx950 HIP layer.
The >30 s sliding-window split and merge (S3Tokenizer.swift:497-650, S3TokenizerUtils.swift:71-88) is integer bookkeeping and
stays on the host, as in the Swift."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .whisper import _TensorView


class _S3Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "n_audio_state", "n_audio_head", "n_audio_layer")]


def _declare(lib):
    if getattr(lib, "_s3_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.mia_s3tok_load.restype = vp
    lib.mia_s3tok_load.argtypes = [vp, C.POINTER(_S3Cfg), C.POINTER(_TensorView), i32]
    lib.mia_s3tok_free.restype = None
    lib.mia_s3tok_free.argtypes = [vp]
    lib.mia_s3tok_encode.restype = i32
    lib.mia_s3tok_encode.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, i32]
    lib._s3_declared = True


def merge_tokenized_segments(segments, overlap=4, token_rate=25):
    out = []
    ot = (overlap // 2) * token_rate
    for i, toks in enumerate(segments):
        left = 0 if i == 0 else ot
        right = len(toks) - ot if i != len(segments) - 1 else len(toks)
        if left < right:
            out.extend(toks[left:right])
    return out


class S3Tokenizer:
    def __init__(self, ctx, h, cfg):
        self.ctx, self.h, self.cfg = ctx, h, cfg
        ctx.adopt(self)

    @staticmethod
    def load(ctx: _lib.Context, cfg, weights: dict[str, np.ndarray]) -> "S3Tokenizer":
        _declare(ctx.lib)
        c = _S3Cfg(cfg.n_mels, cfg.n_audio_state, cfg.n_audio_head, cfg.n_audio_layer)
        views = (_TensorView * len(weights))()
        keep = []
        for i, (name, arr) in enumerate(weights.items()):
            a = np.ascontiguousarray(arr, np.float32)
            keep.append(a)
            views[i] = _TensorView(name.encode(), _lib.F32, a.ndim, (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim))), a.ctypes.data)
        h = ctx.lib.mia_s3tok_load(ctx.h, C.byref(c), views, len(weights))
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return S3Tokenizer(ctx, h, cfg)

    def close(self):
        if self.h and getattr(self.ctx, 'h', None):
            self.ctx.lib.mia_s3tok_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _encode(self, mel: np.ndarray, mel_len: np.ndarray):
        B, _, T = mel.shape
        stride = ((T - 1) // 2 + 1 - 1) // 2 + 1
        toks = np.zeros((B, stride), np.int32)
        n = np.zeros(B, np.int32)
        ml = np.ascontiguousarray(mel_len, np.int32)
        m = np.ascontiguousarray(mel, np.float32)
        self.ctx.check(self.ctx.lib.mia_s3tok_encode(self.h, m.ctypes.data, ml.ctypes.data, B, T, toks.ctypes.data, stride, n.ctypes.data, _lib.MEM_HOST))
        return toks, n

    def quantize(self, mel: np.ndarray, mel_len) -> tuple[np.ndarray, np.ndarray]:
        """quantize(mel:melLen:): mel float32 [B, n_mels, T]; clips above 3000 frames go through the 30 s / 4 s-overlap windows."""
        mel_len = np.asarray(mel_len, np.int64)
        if (mel_len <= 3000).all():
            return self._encode(mel, mel_len)
        outs = []
        for b in range(mel.shape[0]):
            L = int(mel_len[b])
            if L <= 3000:
                t, n = self._encode(mel[b:b + 1, :, :L], [L])
                outs.append(t[0, :n[0]].tolist())
                continue
            segs, start = [], 0
            while start < L:                                  # S3Tokenizer.swift:526-560: windows of 3000 frames, stride 2600
                end = min(start + 3000, L)
                t, n = self._encode(np.ascontiguousarray(mel[b:b + 1, :, start:end]), [end - start])
                segs.append(t[0, :n[0]].tolist())
                if end == L:
                    break
                start += 2600
            outs.append(merge_tokenized_segments(segs))
        n = np.asarray([len(o) for o in outs], np.int32)
        toks = np.zeros((len(outs), int(n.max())), np.int32)
        for b, o in enumerate(outs):
            toks[b, :len(o)] = o
        return toks, n
