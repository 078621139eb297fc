"""Operator-level host wrappers (include/mia.h "operator level"): used by kernel-level parity tests and micro-benchmarks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import audio as _audio


def _declare(lib):
    if getattr(lib, "_ops_declared", False):
        return
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.mia_op_linear.restype = i32
    lib.mia_op_linear.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32]
    lib._ops_declared = True


def to16(a: np.ndarray, dtype: int) -> np.ndarray:
    return np.ascontiguousarray(a, np.float32).astype(np.float16) if dtype == _lib.F16 else _audio.f32_to_bf16(a)


def from16(a: np.ndarray, dtype: int) -> np.ndarray:
    return a.astype(np.float32) if dtype == _lib.F16 else _audio.bf16_to_f32(a)


def linear(ctx: _lib.Context, x: np.ndarray, w: np.ndarray, bias=None, residual=None, act: str | None = None,
           dtype: int = _lib.BF16, out_f32: bool = False, variant: int = 1) -> np.ndarray:
    """y = act(x @ w.T + bias) + residual on the GPU; x [M,K], w [N,K] fp32 (rounded to `dtype` here)."""
    _declare(ctx.lib)
    M, K = x.shape
    N = w.shape[0]
    x16, w16 = to16(x, dtype), to16(w, dtype)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if residual is None else np.ascontiguousarray(residual, np.float32)
    y = np.empty((M, N), np.float32 if out_f32 else (np.float16 if dtype == _lib.F16 else np.uint16))
    ctx.check(ctx.lib.mia_op_linear(ctx.h, x16.ctypes.data, K, w16.ctypes.data, None if b is None else b.ctypes.data,
                                    None if r is None else r.ctypes.data, N, y.ctypes.data, N, M, N, K,
                                    1 if act == "gelu" else 0, dtype, 1 if out_f32 else 0, variant, _lib.MEM_HOST))
    return y if out_f32 else from16(y, dtype)
