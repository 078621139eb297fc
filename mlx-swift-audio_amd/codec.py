"""Host-side mirror of the reference's codec-decoder interface, backed by the gfx950 HIP layer (include/mia.h):
SNACDecoder.decode(codes:) (TTS/Orpheus/SNAC/SNACDecoder.swift:281-289) and DACCodec.decodeFromCodes
(Codec/DAC/DACModel.swift:303-306).  No arithmetic happens in Python."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .whisper import _TensorView


class _SnacCfg(C.Structure):
    _fields_ = [("latent_dim", C.c_int32), ("decoder_dim", C.c_int32), ("n_rates", C.c_int32), ("decoder_rates", C.c_int32 * 8),
                ("n_vq", C.c_int32), ("vq_strides", C.c_int32 * 4), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
                ("noise", C.c_int32), ("depthwise", C.c_int32)]


class _DacCfg(C.Structure):
    _fields_ = [("latent_dim", C.c_int32), ("decoder_dim", C.c_int32), ("n_rates", C.c_int32), ("decoder_rates", C.c_int32 * 8),
                ("n_codebooks", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32)]


class _DacEncCfg(C.Structure):
    _fields_ = [("encoder_dim", C.c_int32), ("n_rates", C.c_int32), ("encoder_rates", C.c_int32 * 8)]


def _declare(lib):
    if getattr(lib, "_codec_declared", False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.mia_dac_load_encoder.restype = i32
    lib.mia_dac_load_encoder.argtypes = [vp, C.POINTER(_DacEncCfg), C.POINTER(_TensorView), i32]
    lib.mia_dac_code_len.restype = i64
    lib.mia_dac_code_len.argtypes = [vp, i64]
    lib.mia_dac_encode.restype = i32
    lib.mia_dac_encode.argtypes = [vp, vp, i64, i32, vp, i64, C.POINTER(i64), i32]
    lib.mia_snac_load.restype = vp
    lib.mia_snac_load.argtypes = [vp, C.POINTER(_SnacCfg), C.POINTER(_TensorView), i32]
    lib.mia_dac_load.restype = vp
    lib.mia_dac_load.argtypes = [vp, C.POINTER(_DacCfg), C.POINTER(_TensorView), i32]
    lib.mia_codec_free.restype = None
    lib.mia_codec_free.argtypes = [vp]
    lib.mia_codec_output_len.restype = i64
    lib.mia_codec_output_len.argtypes = [vp, i64]
    lib.mia_codec_noise_len.restype = i64
    lib.mia_codec_noise_len.argtypes = [vp, i64]
    lib.mia_snac_decode.restype = i32
    lib.mia_snac_decode.argtypes = [vp, C.POINTER(vp), vp, i32, vp, i64, vp, i64, C.POINTER(i64), i32]
    lib.mia_dac_decode.restype = i32
    lib.mia_dac_decode.argtypes = [vp, vp, i32, i64, vp, i64, C.POINTER(i64), i32]
    lib._codec_declared = True


def _views(weights):
    views = (_TensorView * len(weights))()
    keep = []
    for i, (name, arr) in enumerate(weights.items()):
        a = np.ascontiguousarray(arr, np.float32)
        keep.append(a)
        shp = (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim)))
        views[i] = _TensorView(name.encode(), _lib.F32, a.ndim, shp, a.ctypes.data)
    return views, keep


class _Codec:
    def __init__(self, ctx, h, cfg):
        self.ctx, self.h, self.cfg = ctx, h, cfg
        ctx.adopt(self)

    def output_len(self, latent_len: int) -> int:
        return int(self.ctx.lib.mia_codec_output_len(self.h, latent_len))

    def noise_len(self, latent_len: int) -> int:
        return int(self.ctx.lib.mia_codec_noise_len(self.h, latent_len))

    def close(self):
        if self.h and getattr(self.ctx, 'h', None):
            self.ctx.lib.mia_codec_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SNACDecoder(_Codec):
    @staticmethod
    def load(ctx: _lib.Context, cfg, weights: dict[str, np.ndarray]) -> "SNACDecoder":
        _declare(ctx.lib)
        c = _SnacCfg(cfg.latent_dim, cfg.decoder_dim, len(cfg.decoder_rates), (C.c_int32 * 8)(*cfg.decoder_rates), len(cfg.vq_strides),
                     (C.c_int32 * 4)(*cfg.vq_strides), cfg.codebook_size, cfg.codebook_dim, 1 if cfg.noise else 0, 1 if cfg.depthwise else 0)
        views, keep = _views(weights)
        h = ctx.lib.mia_snac_load(ctx.h, C.byref(c), views, len(weights))
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return SNACDecoder(ctx, h, cfg)

    def decode(self, codes: list[list[int]], noise: np.ndarray | None = None) -> np.ndarray:
        """decode(codes:) -> float32 [samples]; `noise` = explicit N(0,1) draws for the NoiseBlocks (None = none)."""
        arrs = [np.ascontiguousarray(c, np.int32) for c in codes]
        n = np.asarray([a.size for a in arrs], np.int32)
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        T0 = max(int(a.size) * s for a, s in zip(arrs, self.cfg.vq_strides))
        pcm = np.empty(self.output_len(T0), np.float32)
        ns = C.c_int64(0)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
        self.ctx.check(self.ctx.lib.mia_snac_decode(self.h, ptrs, n.ctypes.data, len(arrs), None if nz is None else nz.ctypes.data,
                                                    0 if nz is None else nz.size, pcm.ctypes.data, pcm.size, C.byref(ns), _lib.MEM_HOST))
        return pcm[:ns.value]


class DACCodec(_Codec):
    @staticmethod
    def load(ctx: _lib.Context, cfg, weights: dict[str, np.ndarray]) -> "DACCodec":
        _declare(ctx.lib)
        c = _DacCfg(cfg.latent_dim, cfg.decoder_dim, len(cfg.decoder_rates), (C.c_int32 * 8)(*cfg.decoder_rates), cfg.n_codebooks,
                    cfg.codebook_size, cfg.codebook_dim)
        views, keep = _views(weights)
        h = ctx.lib.mia_dac_load(ctx.h, C.byref(c), views, len(weights))
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        codec = DACCodec(ctx, h, cfg)
        if "encoder.block.layers.0.weight_v" in weights:            # checkpoints carry the encoder; decode-only callers may drop it
            e = _DacEncCfg(cfg.encoder_dim, len(cfg.encoder_rates), (C.c_int32 * 8)(*cfg.encoder_rates))
            ctx.check(ctx.lib.mia_dac_load_encoder(h, C.byref(e), views, len(weights)))
        return codec

    def encode(self, audio: np.ndarray, n_quantizers: int | None = None) -> np.ndarray:
        """DACCodec.encode(_:nQuantizers:) for one mono sequence (DACModel.swift:284-296): float32 [samples] -> codes int32 [n_q, T]."""
        a = np.ascontiguousarray(audio, np.float32).reshape(-1)
        T = int(self.ctx.lib.mia_dac_code_len(self.h, a.size))
        nq = self.cfg.n_codebooks if not n_quantizers else min(n_quantizers, self.cfg.n_codebooks)
        codes = np.zeros((nq, max(T, 1)), np.int32)
        ns = C.c_int64(0)
        self.ctx.check(self.ctx.lib.mia_dac_encode(self.h, a.ctypes.data, a.size, nq, codes.ctypes.data, codes.shape[1], C.byref(ns), _lib.MEM_HOST))
        return codes[:, :ns.value]

    def decode_from_codes(self, codes: np.ndarray) -> np.ndarray:
        """decodeFromCodes: codes int [B, n_codebooks, T] (or [n_codebooks, T]) -> float32 [B, samples] (or [samples])."""
        codes = np.ascontiguousarray(codes, np.int32)
        single = codes.ndim == 2
        if single:
            codes = codes[None]
        B, ncb, T = codes.shape
        out = np.empty((B, self.output_len(T)), np.float32)
        for b in range(B):
            ns = C.c_int64(0)
            self.ctx.check(self.ctx.lib.mia_dac_decode(self.h, codes[b].ctypes.data, ncb, T, out[b].ctypes.data, out.shape[1], C.byref(ns), _lib.MEM_HOST))
        return out[0] if single else out
