"""Host-side mirror of CosyVoice2FlowModule (TTS/CosyVoice2/CosyVoice2Model.swift:401-553), backed by the gfx950 HIP layer.
`inference` keeps the Swift method's argument names; the CFM's initial noise `z` is explicit (the reference draws it internally)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .whisper import _TensorView


class _FlowCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_size", "output_size", "spk_embed_dim", "vocab_size", "pre_lookahead_len", "n_timesteps",
                                         "enc_heads", "enc_linear_units", "enc_blocks", "enc_up_blocks", "upsample_stride",
                                         "dec_in_channels", "dec_channels", "dec_heads", "dec_n_blocks", "dec_mid_blocks")] + [("cfg_rate", C.c_float)]


def _declare(lib):
    if getattr(lib, "_flow_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.mia_flow_load.restype = vp
    lib.mia_flow_load.argtypes = [vp, C.POINTER(_FlowCfg), C.POINTER(_TensorView), i32]
    lib.mia_flow_free.restype = None
    lib.mia_flow_free.argtypes = [vp]
    lib.mia_flow_encode.restype = i32
    lib.mia_flow_encode.argtypes = [vp, vp, i32, vp, i32]
    lib.mia_flow_inference.restype = i32
    lib.mia_flow_inference.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32]
    lib._flow_declared = True


class FlowModule:
    def __init__(self, ctx, h, cfg):
        self.ctx, self.h, self.cfg = ctx, h, cfg
        ctx.adopt(self)

    @staticmethod
    def load(ctx: _lib.Context, cfg, weights: dict[str, np.ndarray]) -> "FlowModule":
        _declare(ctx.lib)
        c = _FlowCfg(cfg.input_size, cfg.output_size, cfg.spk_embed_dim, cfg.vocab_size, cfg.pre_lookahead_len, cfg.n_timesteps,
                     cfg.enc_heads, cfg.enc_linear_units, cfg.enc_blocks, cfg.enc_up_blocks, cfg.upsample_stride,
                     cfg.dec_in_channels, cfg.dec_channels, cfg.dec_heads, cfg.dec_n_blocks, cfg.dec_mid_blocks, cfg.cfg_rate)
        views = (_TensorView * len(weights))()
        keep = []
        for i, (name, arr) in enumerate(weights.items()):
            a = np.ascontiguousarray(arr, np.float32)
            keep.append(a)
            views[i] = _TensorView(name.encode(), _lib.F32, a.ndim, (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim))), a.ctypes.data)
        h = ctx.lib.mia_flow_load(ctx.h, C.byref(c), views, len(weights))
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return FlowModule(ctx, h, cfg)

    def close(self):
        if self.h and getattr(self.ctx, 'h', None):
            self.ctx.lib.mia_flow_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode(self, token: np.ndarray) -> np.ndarray:
        """encoder_proj(encoder(input_embedding(token))) -> mu [2 n, 80]"""
        t = np.ascontiguousarray(token, np.int32)
        mu = np.empty((t.shape[0] * self.cfg.upsample_stride, self.cfg.output_size), np.float32)
        self.ctx.check(self.ctx.lib.mia_flow_encode(self.h, t.ctypes.data, t.shape[0], mu.ctypes.data, _lib.MEM_HOST))
        return mu

    def inference_streaming(self, token, prompt_token, prompt_feat, embedding, z, n_timesteps: int | None = None, finalize: bool = True,
                            enc_static_chunk: int = 0, dec_static_chunk: int = 0) -> np.ndarray:
        """inference(...) with the modules' chunked-synthesis switches (mia_flow_inference_streaming): finalize = False trims the encoder's
        look-ahead frames (z is then [80, T - pre_lookahead_len * upsample_stride]); *_static_chunk > 0 = streaming attention masks."""
        lib = self.ctx.lib
        lib.mia_flow_inference_streaming.restype = C.c_int
        lib.mia_flow_inference_streaming.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                     C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int]
        t = np.ascontiguousarray(token, np.int32)
        pt = np.ascontiguousarray(prompt_token, np.int32)
        pf = np.ascontiguousarray(prompt_feat, np.float32).reshape(-1, self.cfg.output_size)
        e = np.ascontiguousarray(embedding, np.float32).reshape(-1)
        zz = np.ascontiguousarray(z, np.float32)
        T = (t.shape[0] + pt.shape[0]) * self.cfg.upsample_stride
        trim = 0 if finalize else self.cfg.pre_lookahead_len * self.cfg.upsample_stride
        T = T - trim if T > trim else T
        if zz.shape != (self.cfg.output_size, T) or e.shape[0] != self.cfg.spk_embed_dim:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"flow: z must be [{self.cfg.output_size}, {T}] and embedding [{self.cfg.spk_embed_dim}]")
        mel = np.empty((self.cfg.output_size, T - pf.shape[0]), np.float32)
        nf = C.c_int(0)
        self.ctx.check(lib.mia_flow_inference_streaming(self.h, t.ctypes.data, t.shape[0], pt.ctypes.data if pt.size else None, pt.shape[0],
                                                        pf.ctypes.data if pf.size else None, pf.shape[0], e.ctypes.data, zz.ctypes.data, n_timesteps or 0,
                                                        1 if finalize else 0, enc_static_chunk, dec_static_chunk, mel.ctypes.data, C.byref(nf), _lib.MEM_HOST))
        assert nf.value == mel.shape[1]
        return mel

    def inference_batch(self, utterances, n_timesteps: int | None = None) -> list:
        """Several utterances through one pass of the flow (mia_flow_inference_batch).  `utterances`: sequence of
        (token, prompt_token, prompt_feat, embedding, z) tuples with the shapes inference() takes; returns their mels in order, each
        bit-identical to its own inference() call."""
        lib = self.ctx.lib
        lib.mia_flow_inference_batch.restype = C.c_int
        lib.mia_flow_inference_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_int]
        U = len(utterances)
        keep, mels = [], []
        ptr = lambda n: (C.c_void_p * n)()
        tok, ptok, pfs, embs, zs, outs = ptr(U), ptr(U), ptr(U), ptr(U), ptr(U), ptr(U)
        n_tok, n_pt, n_pf = np.zeros(U, np.int32), np.zeros(U, np.int32), np.zeros(U, np.int32)
        for i, (token, prompt_token, prompt_feat, embedding, z) in enumerate(utterances):
            t = np.ascontiguousarray(token, np.int32)
            pt = np.ascontiguousarray(prompt_token, np.int32)
            pf = np.ascontiguousarray(prompt_feat, np.float32).reshape(-1, self.cfg.output_size)
            e = np.ascontiguousarray(embedding, np.float32).reshape(-1)
            zz = np.ascontiguousarray(z, np.float32)
            T = (t.shape[0] + pt.shape[0]) * self.cfg.upsample_stride
            if zz.shape != (self.cfg.output_size, T) or e.shape[0] != self.cfg.spk_embed_dim:
                raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"flow: utterance {i}: z must be [{self.cfg.output_size}, {T}] and embedding [{self.cfg.spk_embed_dim}]")
            mel = np.empty((self.cfg.output_size, T - pf.shape[0]), np.float32)
            keep += [t, pt, pf, e, zz]
            mels.append(mel)
            tok[i], ptok[i], pfs[i] = t.ctypes.data, (pt.ctypes.data if pt.size else None), (pf.ctypes.data if pf.size else None)
            embs[i], zs[i], outs[i] = e.ctypes.data, zz.ctypes.data, mel.ctypes.data
            n_tok[i], n_pt[i], n_pf[i] = t.shape[0], pt.shape[0], pf.shape[0]
        self.ctx.check(lib.mia_flow_inference_batch(self.h, U, tok, n_tok.ctypes.data, ptok, n_pt.ctypes.data, pfs, n_pf.ctypes.data, embs, zs,
                                                    n_timesteps or 0, outs, _lib.MEM_HOST))
        return mels

    def inference(self, token, prompt_token, prompt_feat, embedding, z, n_timesteps: int | None = None) -> np.ndarray:
        """token [n], prompt_token [m] (may be empty), prompt_feat [m1, 80], embedding [spk_embed_dim], z [80, 2 (n + m)] -> mel [80, 2 (n + m) - m1]"""
        t = np.ascontiguousarray(token, np.int32)
        pt = np.ascontiguousarray(prompt_token, np.int32)
        pf = np.ascontiguousarray(prompt_feat, np.float32).reshape(-1, self.cfg.output_size)
        e = np.ascontiguousarray(embedding, np.float32).reshape(-1)
        zz = np.ascontiguousarray(z, np.float32)
        T = (t.shape[0] + pt.shape[0]) * self.cfg.upsample_stride
        if zz.shape != (self.cfg.output_size, T) or e.shape[0] != self.cfg.spk_embed_dim:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"flow: z must be [{self.cfg.output_size}, {T}] and embedding [{self.cfg.spk_embed_dim}]")
        mel = np.empty((self.cfg.output_size, T - pf.shape[0]), np.float32)
        self.ctx.check(self.ctx.lib.mia_flow_inference(self.h, t.ctypes.data, t.shape[0], pt.ctypes.data if pt.size else None, pt.shape[0],
                                                       pf.ctypes.data if pf.size else None, pf.shape[0], e.ctypes.data, zz.ctypes.data,
                                                       n_timesteps or 0, mel.ctypes.data, _lib.MEM_HOST))
        return mel
