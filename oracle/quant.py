"""TEST INFRASTRUCTURE ONLY -- MLX affine quantisation restated from its public documentation (mx.quantize / mx.dequantize,
mode "affine"): per group of `group_size` consecutive elements along the last axis, w ~ scale * q + bias with q in [0, 2^bits),
scale = (max - min) / (2^bits - 1), bias = min; codes packed little end first into uint32 words.  (MLX nudges scale / bias so that
zero is exactly representable; de-quantisation -- the direction the hot path needs -- does not depend on that detail.)
The reference uses it through `quantize(model:)` (STT/Whisper/WhisperModel.swift:189-196).  PARITY UNPINNED (no MLX here)."""
from __future__ import annotations

import numpy as np


def quantize_affine(w: np.ndarray, group_size: int = 64, bits: int = 4):
    """w float [..., cols] -> (packed uint32 [..., cols * bits / 32], scales, biases [..., cols / group_size]) (float32)."""
    w = np.asarray(w, np.float32)
    *lead, cols = w.shape
    g = w.reshape(-1, cols // group_size, group_size)
    lo, hi = g.min(-1), g.max(-1)
    n = float((1 << bits) - 1)
    scale = np.where(hi > lo, (hi - lo) / n, 1.0).astype(np.float32)
    q = np.clip(np.rint((g - lo[..., None]) / scale[..., None]), 0, n).astype(np.uint32)
    per = 32 // bits
    q = q.reshape(-1, cols // per, per)
    packed = np.zeros(q.shape[:2], np.uint32)
    for j in range(per):
        packed |= q[:, :, j] << np.uint32(j * bits)
    return packed.reshape(*lead, cols // per), scale.reshape(*lead, -1), lo.astype(np.float32).reshape(*lead, -1)


def dequantize_affine(packed: np.ndarray, scales: np.ndarray, biases: np.ndarray, group_size: int = 64, bits: int = 4) -> np.ndarray:
    packed = np.asarray(packed, np.uint32)
    per = 32 // bits
    *lead, words = packed.shape
    cols = words * per
    shifts = (np.arange(per, dtype=np.uint32) * np.uint32(bits))
    q = ((packed[..., None] >> shifts) & np.uint32((1 << bits) - 1)).reshape(*lead, cols).astype(np.float32)
    s = np.repeat(np.asarray(scales, np.float32), group_size, axis=-1)
    b = np.repeat(np.asarray(biases, np.float32), group_size, axis=-1)
    return (s * q + b).astype(np.float32)
