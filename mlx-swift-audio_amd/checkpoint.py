"""On-disk formats of the reference's checkpoints: a safetensors reader, `config.json` -> ModelDimensions,
and the expansion of MLX affine-quantised layers (`<path>.weight` uint32 + `.scales` + `.biases`, group 64, 4 / 8 bit -- the
reference's DEFAULT checkpoints, STT/Whisper/WhisperModel.swift:144-206) into the dense tensors the loaders of this package take.
De-quantisation runs on the GPU (mia_dequant_affine); nothing here falls back to the CPU."""
from __future__ import annotations

import ctypes as C
import json
import mmap
import struct

import numpy as np

from . import _lib

_DT = {"F32": (np.float32, 4), "F16": (np.float16, 2), "BF16": (np.uint16, 2), "I32": (np.int32, 4), "U32": (np.uint32, 4), "I64": (np.int64, 8),
       "U8": (np.uint8, 1), "I8": (np.int8, 1), "U16": (np.uint16, 2), "I16": (np.int16, 2), "F64": (np.float64, 8), "BOOL": (np.bool_, 1)}


class Tensor(np.ndarray):
    """ndarray view into the mapped file; `.st_dtype` keeps the safetensors dtype tag (bf16 payloads are exposed as uint16)."""
    st_dtype = "F32"


def read_safetensors(path: str) -> dict[str, np.ndarray]:
    """Zero-copy reader of the safetensors container: u64 header length, JSON header {name: {dtype, shape, data_offsets}}, raw little-endian
    tensors.  (What MLX.loadArrays(url:) does at WhisperModel.swift:184.)"""
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    (n,) = struct.unpack("<Q", mm[:8])
    header = json.loads(mm[8:8 + n].decode("utf-8"))
    base = 8 + n
    out: dict[str, np.ndarray] = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        if meta["dtype"] not in _DT:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"safetensors: unsupported dtype {meta['dtype']} for '{name}'")
        dt, sz = _DT[meta["dtype"]]
        b0, b1 = meta["data_offsets"]
        count = int(np.prod(meta["shape"], dtype=np.int64)) if meta["shape"] else 1
        if b1 - b0 != count * sz:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"safetensors: '{name}' has {b1 - b0} bytes for shape {meta['shape']}")
        a = np.frombuffer(mm, dtype=dt, count=count, offset=base + b0).reshape(meta["shape"]).view(Tensor)
        a.st_dtype = meta["dtype"]
        out[name] = a
    return out


def bf16_bits_to_f32(a: np.ndarray) -> np.ndarray:
    return (np.asarray(a, np.uint16).astype(np.uint32) << 16).view(np.float32)


def to_float32(a: np.ndarray) -> np.ndarray:
    if getattr(a, "st_dtype", None) == "BF16":
        return bf16_bits_to_f32(a)
    return np.asarray(a, np.float32)


def load_model_dimensions(config_path: str):
    """ModelDimensions.load(from:) (STT/Whisper/Config/WhisperConfig.swift:78-86): the ten n_* keys of config.json."""
    from .synthetic import ModelDimensions
    cfg = json.load(open(config_path))
    keys = ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")
    missing = [k for k in keys if k not in cfg]
    if missing:
        raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"config.json lacks {missing}")
    return ModelDimensions(*[int(cfg[k]) for k in keys])


def _mia_dtype(a: np.ndarray) -> int:
    tag = getattr(a, "st_dtype", None)
    if tag == "BF16":
        return _lib.BF16
    if a.dtype == np.float16:
        return _lib.F16
    if a.dtype == np.float32:
        return _lib.F32
    raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"scales / biases must be f32, f16 or bf16 (got {a.dtype})")


def dequantize_affine(ctx: _lib.Context, wq: np.ndarray, scales: np.ndarray, biases: np.ndarray, group_size: int = 64, bits: int = 4) -> np.ndarray:
    """GPU expansion of one quantised tensor to float32 [rows, cols]."""
    lib = ctx.lib
    if not getattr(lib, "_dq_declared", False):
        lib.mia_dequant_affine.restype = C.c_int
        lib.mia_dequant_affine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        lib._dq_declared = True
    w = np.ascontiguousarray(wq).view(np.uint32) if wq.dtype != np.uint32 else np.ascontiguousarray(wq)
    rows = int(np.prod(w.shape[:-1]))
    cols = w.shape[-1] * (32 // bits)
    sdt = _mia_dtype(scales)
    s, b = np.ascontiguousarray(scales), np.ascontiguousarray(biases)
    if s.size != rows * (cols // group_size) or b.size != s.size or _mia_dtype(biases) != sdt:
        raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "dequantize_affine: scales / biases do not match the packed weight")
    out = np.empty(tuple(w.shape[:-1]) + (cols,), np.float32)
    ctx.check(lib.mia_dequant_affine(ctx.h, w.ctypes.data, s.ctypes.data, b.ctypes.data, rows, cols, group_size, bits, sdt, out.ctypes.data, _lib.F32, _lib.MEM_HOST))
    return out


def expand_checkpoint(ctx: _lib.Context, tensors: dict[str, np.ndarray], bits: int | None = None, group_size: int = 64) -> dict[str, np.ndarray]:
    """Dense float32 tensors under the Module key schema: every `<p>.weight` that comes with `<p>.scales` is de-quantised (bits inferred
    from the packed width when not given), `.scales` / `.biases` are dropped, bf16 payloads become float32."""
    out: dict[str, np.ndarray] = {}
    for name, a in tensors.items():
        if name.endswith(".scales") or (name.endswith(".biases") and name[:-7] + ".scales" in tensors):
            continue
        if name.endswith(".weight") and name[:-7] + ".scales" in tensors:
            sc, bi = tensors[name[:-7] + ".scales"], tensors[name[:-7] + ".biases"]
            b = bits or (32 * a.shape[-1]) // (sc.shape[-1] * group_size)
            out[name] = dequantize_affine(ctx, a, sc, bi, group_size, b)
        else:
            out[name] = to_float32(a) if a.dtype.kind in "fu" and getattr(a, "st_dtype", "F32") in ("F32", "F16", "BF16") else np.asarray(a)
    return out


def load_whisper_checkpoint(ctx: _lib.Context, model_dir: str):
    """WhisperModel.load's file half (WhisperModel.swift:175-206): config.json + model.safetensors -> (ModelDimensions, dense tensors)."""
    import os
    dims = load_model_dimensions(os.path.join(model_dir, "config.json"))
    return dims, expand_checkpoint(ctx, read_safetensors(os.path.join(model_dir, "model.safetensors")))


def quantize_affine(w: np.ndarray, group_size: int = 64, bits: int = 4, scale_dtype=np.float16):
    """A SIMPLIFIED min / max affine quantiser in MLX's storage layout -- the direction the reference's loaders take for checkpoints
    that are not already quantised (`quantize(model:) { (64, bits, .affine) }`, STT/Whisper/WhisperModel.swift:189-196): per group of
    `group_size` consecutive inputs, scale = (max - min) / (2^bits - 1), bias = min, code = round((w - bias) / scale); codes packed
    little end first into uint32 words.  mx.quantize additionally nudges scale / bias (edge and sign handling), which this does not
    reproduce: parity of the CODES with mx.quantize is unpinned; the contract of the hot path is de-quantisation (scale * code + bias),
    which is layout-exact.  Returns (codes uint32 [N, K*bits/32], scales, biases [N, K/group_size] in `scale_dtype`) -- the three
    tensors a quantised checkpoint stores for the layer (feed them to CausalLM.attach_q4 / expand them with dequantize_affine)."""
    w = np.asarray(w, np.float32)
    n, k = w.shape
    g = w.reshape(n, k // group_size, group_size)
    lo, hi = g.min(-1), g.max(-1)
    top = float((1 << bits) - 1)
    scale = np.where(hi > lo, (hi - lo) / top, 1.0).astype(scale_dtype)
    bias = lo.astype(scale_dtype)
    s32 = scale.astype(np.float32)[..., None]
    q = np.clip(np.rint((g - bias.astype(np.float32)[..., None]) / s32), 0, top).astype(np.uint32)
    per = 32 // bits
    q = q.reshape(n, k // per, per)
    packed = np.zeros(q.shape[:2], np.uint32)
    for j in range(per):
        packed |= q[:, :, j] << np.uint32(j * bits)
    return packed, scale, bias
