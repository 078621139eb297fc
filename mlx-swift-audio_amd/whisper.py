"""Host-side mirror of the reference's Whisper interface (WhisperModel / GreedyDecoder / DecodingOptions),
backed by the gfx950 HIP layer through the C ABI (include/mia.h).  No arithmetic happens in Python.

Reference interface mirrored (paths relative to /root/reference/package/STT/Whisper):
  WhisperModel.load / encode / decode / detectLanguage      WhisperModel.swift:59-76,144-214,223-260
  DecodingOptions / DecodingResult / GreedyDecoder.decode   WhisperDecoding.swift:14-74,96-389
  special-token arithmetic, sotSequence                     WhisperTokenizer.swift:72-96,377-396
The tokenizer (a CPU text codec needing the tiktoken vocabulary) stays outside: its integer outputs (suppress list,
blank ids, prompt ids) are passed in, exactly the arrays GreedyDecoder asks the tokenizer for.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from . import audio as _audio


class _Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                         "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class _TensorView(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("data", C.c_void_p)]


class _DecodeOpts(C.Structure):
    _fields_ = [("initial_tokens", C.c_void_p), ("n_initial", C.c_int32), ("per_clip_initial", C.c_int32),
                ("sot_index", C.c_int32), ("suppress_ids", C.c_void_p), ("n_suppress", C.c_int32),
                ("blank_ids", C.c_void_p), ("n_blank", C.c_int32),
                ("eot", C.c_int32), ("no_speech", C.c_int32), ("no_timestamps", C.c_int32), ("timestamp_begin", C.c_int32),
                ("timestamps", C.c_int32), ("max_tokens", C.c_int32), ("max_initial_timestamp_index", C.c_int32),
                ("max_new_tokens", C.c_int32), ("temperature", C.c_float), ("uniforms", C.c_void_p),
                ("n_initial_per_clip", C.c_void_p), ("sot_index_per_clip", C.c_void_p), ("clip_temperature", C.c_void_p),
                ("clip_active", C.c_void_p)]


def _declare(lib):
    if getattr(lib, "_whisper_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.mia_whisper_load.restype = vp
    lib.mia_whisper_load.argtypes = [vp, C.POINTER(_Dims), C.POINTER(_TensorView), i32, i32]
    lib.mia_whisper_free.restype = None
    lib.mia_whisper_free.argtypes = [vp]
    lib.mia_whisper_encode.restype = i32
    lib.mia_whisper_encode.argtypes = [vp, vp, i32, i32]
    lib.mia_whisper_get_audio_features.restype = i32
    lib.mia_whisper_get_audio_features.argtypes = [vp, vp, i32, i32]
    lib.mia_whisper_decode_greedy.restype = i32
    lib.mia_whisper_decode_greedy.argtypes = [vp, C.POINTER(_DecodeOpts), vp, vp, vp, vp, i32]
    lib.mia_whisper_detect_language.restype = i32
    lib.mia_whisper_detect_language.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    lib.mia_whisper_encode_windows.restype = i32
    lib.mia_whisper_encode_windows.argtypes = [vp, vp, vp, i32, C.c_int64, i32]
    lib.mia_whisper_transcribe_windows.restype = i32
    lib.mia_whisper_transcribe_windows.argtypes = [vp, vp, vp, i32, C.c_int64, C.POINTER(_DecodeOpts), vp, vp, vp, vp, i32]
    lib._whisper_declared = True


@dataclass
class SpecialTokens:
    """WhisperTokenizer.swift:72-96: ids follow from n_vocab alone."""
    eot: int
    sot: int
    translate: int
    transcribe: int
    sot_lm: int
    sot_prev: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int
    is_multilingual: bool
    num_languages: int

    @staticmethod
    def for_vocab(n_vocab: int) -> "SpecialTokens":
        multilingual = n_vocab >= 51865
        num_languages = n_vocab - 51765 - (1 if multilingual else 0)
        nxt = 50257 if multilingual else 50256
        ids = []
        for skip in (0, 0, num_languages, 0, 0, 0, 0, 0):
            nxt += skip
            ids.append(nxt)
            nxt += 1
        eot, sot, translate, transcribe, sot_lm, sot_prev, no_speech, no_timestamps = ids
        return SpecialTokens(eot, sot, translate, transcribe, sot_lm, sot_prev, no_speech, no_timestamps, nxt,
                             multilingual, num_languages)

    def sot_sequence(self, language_index: int | None = 0, task: str = "transcribe") -> list[int]:
        seq = [self.sot]
        if not self.is_multilingual:
            return seq
        if language_index is not None:
            seq.append(self.sot + 1 + language_index)
        seq.append(self.transcribe if task == "transcribe" else self.translate)
        return seq


@dataclass
class DecodingOptions:
    task: str = "transcribe"
    language_index: int | None = 0
    temperature: float = 0.0
    max_tokens: int = 448
    timestamps: bool = True
    prompt: list[int] = field(default_factory=list)
    suppress_ids: list[int] = field(default_factory=list)
    blank_ids: list[int] = field(default_factory=list)
    max_new_tokens: int = 0
    max_initial_timestamp_index: int = 50


@dataclass
class DecodingResult:
    tokens: list[int]
    avg_logprob: float
    no_speech_prob: float


_NP16 = {_lib.BF16: np.uint16, _lib.F16: np.float16}


class WhisperModel:
    """Device-resident Whisper (weights + batch state)."""

    def __init__(self, ctx: _lib.Context, handle, dims, dtype):
        self.ctx, self.h, self.dims, self.dtype = ctx, handle, dims, dtype
        ctx.adopt(self)
        self.special = SpecialTokens.for_vocab(dims.n_vocab)
        self._keep = []

    @staticmethod
    def load(ctx: _lib.Context, dims, weights: dict[str, np.ndarray], dtype: int = _lib.BF16) -> "WhisperModel":
        """dims: any object with the ModelDimensions fields; weights: name -> fp32/fp16 array with the reference key schema."""
        _declare(ctx.lib)
        cd = _Dims(*[int(getattr(dims, f[0])) for f in _Dims._fields_])
        views = (_TensorView * len(weights))()
        keep = []
        for i, (name, arr) in enumerate(weights.items()):
            if arr.dtype == np.float16:
                a, dt = np.ascontiguousarray(arr), _lib.F16
            else:
                a, dt = np.ascontiguousarray(arr, np.float32), _lib.F32
            keep.append(a)
            shp = (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim)))
            views[i] = _TensorView(name.encode(), dt, a.ndim, shp, a.ctypes.data)
        h = ctx.lib.mia_whisper_load(ctx.h, C.byref(cd), views, len(weights), dtype)
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return WhisperModel(ctx, h, dims, dtype)

    def close(self):
        if self.h and getattr(self.ctx, 'h', None):
            self.ctx.lib.mia_whisper_free(self.h)
            self.h = None

    def set_gemm_variant(self, variant: int) -> None:
        """Test hook: force the encoder GEMM tile variant (0/1: 128^2, 2: 256^2 two-buffer, 3: auto, 4: 256^2 8-phase)."""
        lib = self.ctx.lib
        lib.mia_whisper_set_gemm_variant.restype = C.c_int
        lib.mia_whisper_set_gemm_variant.argtypes = [C.c_void_p, C.c_int]
        self.ctx.check(lib.mia_whisper_set_gemm_variant(self.h, int(variant)))

    def set_debug(self, flags: int) -> None:
        """Test hook (mia_whisper_set_debug): bit 0 = no hipGraph, bit 1 = one-workgroup head."""
        lib = self.ctx.lib
        lib.mia_whisper_set_debug.restype = C.c_int
        lib.mia_whisper_set_debug.argtypes = [C.c_void_p, C.c_int]
        self.ctx.check(lib.mia_whisper_set_debug(self.h, int(flags)))

    def set_weight_sharing(self, concurrent_readers: int) -> None:
        """mia_whisper_set_weight_sharing: > 1 when clones of these weights decode concurrently on other streams (the step then loads
        its weights cacheable); 1 restores the lone-loop (non-temporal) form.  Results are bit-identical."""
        lib = self.ctx.lib
        lib.mia_whisper_set_weight_sharing.restype = C.c_int
        lib.mia_whisper_set_weight_sharing.argtypes = [C.c_void_p, C.c_int]
        self.ctx.check(lib.mia_whisper_set_weight_sharing(self.h, int(concurrent_readers)))

    def set_encode_stream(self, hip_stream: int | None) -> None:
        """mia_whisper_set_encode_stream: run the encoder half of every window on another HIP stream (raw pointer, e.g.
        torch.cuda.Stream(...).cuda_stream); None restores the single-stream form."""
        lib = self.ctx.lib
        lib.mia_whisper_set_encode_stream.restype = C.c_int
        lib.mia_whisper_set_encode_stream.argtypes = [C.c_void_p, C.c_void_p]
        self.ctx.check(lib.mia_whisper_set_encode_stream(self.h, hip_stream))

    def trace_logits(self, clips: list[int]) -> None:
        """Test hook (mia_whisper_trace_logits): keep the raw step logits of these batch rows; [] switches the trace off."""
        lib = self.ctx.lib
        lib.mia_whisper_trace_logits.restype = C.c_int
        lib.mia_whisper_trace_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        a = np.asarray(clips, np.int32)
        self.ctx.check(lib.mia_whisper_trace_logits(self.h, a.ctypes.data if a.size else None, int(a.size)))

    def read_logit_trace(self, slot: int, first_pos: int, n_pos: int) -> np.ndarray:
        """Rows [first_pos, first_pos + n_pos) of trace slot `slot`: float32 [n_pos, n_vocab]."""
        lib = self.ctx.lib
        lib.mia_whisper_read_logit_trace.restype = C.c_int
        lib.mia_whisper_read_logit_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        out = np.empty((n_pos, self.dims.n_vocab), np.float32)
        self.ctx.check(lib.mia_whisper_read_logit_trace(self.h, slot, first_pos, n_pos, out.ctypes.data))
        return out

    def clone(self, ctx: "_lib.Context") -> "WhisperModel":
        """A second handle on the same weights with its own batch state, bound to `ctx` (another stream of the same device)."""
        lib = ctx.lib
        lib.mia_whisper_clone.restype = C.c_void_p
        lib.mia_whisper_clone.argtypes = [C.c_void_p, C.c_void_p]
        h = lib.mia_whisper_clone(self.h, ctx.h)
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, lib.mia_last_error(ctx.h).decode())
        child = WhisperModel(ctx, h, self.dims, self.dtype)
        child._weights_owner = self          # keep the owner alive as long as the clone
        return child

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- encode -------------------------------------------------------------------------------
    def to_compute_dtype(self, mel_f32: np.ndarray) -> np.ndarray:
        """fp32 -> the model's 16-bit storage type (the reference casts the window with .asType(.float16), WhisperSTT.swift:182)."""
        if self.dtype == _lib.F16:
            return np.ascontiguousarray(mel_f32, np.float32).astype(np.float16)
        return _audio.f32_to_bf16(mel_f32)

    def encode(self, mel: np.ndarray) -> None:
        """model.encode(mel): mel [B, 2*n_audio_ctx, n_mels] fp32 (rounded here) or already 16-bit."""
        if mel.ndim == 2:
            mel = mel[None]
        d = self.dims
        if mel.shape[1:] != (2 * d.n_audio_ctx, d.n_mels):
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"mel must be [B,{2 * d.n_audio_ctx},{d.n_mels}], got {mel.shape}")
        m = self.to_compute_dtype(mel) if mel.dtype == np.float32 else np.ascontiguousarray(mel)
        self.B = m.shape[0]
        self.ctx.check(self.ctx.lib.mia_whisper_encode(self.h, m.ctypes.data, self.B, _lib.MEM_HOST))

    def audio_features(self) -> np.ndarray:
        d = self.dims
        out = np.empty((self.B, d.n_audio_ctx, d.n_audio_state), np.float32)
        self.ctx.check(self.ctx.lib.mia_whisper_get_audio_features(self.h, out.ctypes.data, _lib.F32, _lib.MEM_HOST))
        return out

    # ---- decode -------------------------------------------------------------------------------
    def _opts(self, o: DecodingOptions, initial: np.ndarray | None = None):
        st = self.special
        if initial is None:
            toks: list[int] = []
            if o.prompt:
                toks.append(st.sot_prev)
                toks.extend(o.prompt)
            sot_index = len(toks)
            toks.extend(st.sot_sequence(o.language_index, o.task))
            if not o.timestamps:
                toks.append(st.no_timestamps)
            init = np.asarray(toks, np.int32)
            per_clip = 0
        else:
            init = np.ascontiguousarray(initial, np.int32)
            sot_index = int(np.where(init.reshape(-1, init.shape[-1])[0] == st.sot)[0][0])
            per_clip = 1 if init.ndim == 2 else 0
        sup = np.asarray(o.suppress_ids, np.int32)
        blank = np.asarray(o.blank_ids, np.int32)
        keep = (init, sup, blank)
        co = _DecodeOpts(init.ctypes.data, int(init.shape[-1]), per_clip, sot_index,
                         sup.ctypes.data if sup.size else None, int(sup.size), blank.ctypes.data if blank.size else None, int(blank.size),
                         st.eot, st.no_speech, st.no_timestamps, st.timestamp_begin, 1 if o.timestamps else 0, o.max_tokens,
                         o.max_initial_timestamp_index, o.max_new_tokens, float(o.temperature), None, None, None, None, None)
        return co, keep

    def decode_greedy(self, o: DecodingOptions, initial: np.ndarray | None = None) -> list[DecodingResult]:
        """Greedy-decode the B clips of the last encode()."""
        co, keep = self._opts(o, initial)
        B = self.B
        tokens = np.zeros((B, o.max_tokens), np.int32)
        n = np.zeros(B, np.int32)
        avg = np.zeros(B, np.float32)
        nsp = np.zeros(B, np.float32)
        self.ctx.check(self.ctx.lib.mia_whisper_decode_greedy(self.h, C.byref(co), tokens.ctypes.data, n.ctypes.data, avg.ctypes.data,
                                                              nsp.ctypes.data, _lib.MEM_HOST))
        return [DecodingResult(tokens[b, :n[b]].tolist(), float(avg[b]), float(nsp[b])) for b in range(B)]

    def decode_ragged(self, o: DecodingOptions, initials: list[list[int]], sot_index: list[int], temperatures: list[float],
                      uniforms: np.ndarray | None = None, active: list[bool] | None = None) -> list[DecodingResult]:
        """Decode the B clips of the last encode() with per-clip forced prefixes / temperatures (the fallback + prompt
        conditioning cases of WhisperSTT.transcribe).  uniforms [B, max_tokens] feed the T>0 draws."""
        st = self.special
        B = self.B
        assert len(initials) == B and len(sot_index) == B and len(temperatures) == B
        n_max = max(len(t) for t in initials)
        init = np.zeros((B, n_max), np.int32)
        for b, t in enumerate(initials):
            init[b, :len(t)] = t
        n_init = np.asarray([len(t) for t in initials], np.int32)
        sot = np.asarray(sot_index, np.int32)
        temps = np.asarray(temperatures, np.float32)
        act = None if active is None else np.asarray([1 if a else 0 for a in active], np.int32)
        uni = None if uniforms is None else np.ascontiguousarray(uniforms, np.float32)
        if uni is not None and uni.shape != (B, o.max_tokens):
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"uniforms must be [{B},{o.max_tokens}]")
        sup = np.asarray(o.suppress_ids, np.int32)
        blank = np.asarray(o.blank_ids, np.int32)
        co = _DecodeOpts(init.ctypes.data, n_max, 1, int(sot[0]), sup.ctypes.data if sup.size else None, int(sup.size),
                         blank.ctypes.data if blank.size else None, int(blank.size), st.eot, st.no_speech, st.no_timestamps, st.timestamp_begin,
                         1 if o.timestamps else 0, o.max_tokens, o.max_initial_timestamp_index, o.max_new_tokens, 0.0,
                         None if uni is None else uni.ctypes.data, n_init.ctypes.data, sot.ctypes.data, temps.ctypes.data,
                         None if act is None else act.ctypes.data)
        tokens = np.zeros((B, o.max_tokens), np.int32)
        n = np.zeros(B, np.int32)
        avg = np.zeros(B, np.float32)
        nsp = np.zeros(B, np.float32)
        self.ctx.check(self.ctx.lib.mia_whisper_decode_greedy(self.h, C.byref(co), tokens.ctypes.data, n.ctypes.data, avg.ctypes.data,
                                                              nsp.ctypes.data, _lib.MEM_HOST))
        return [DecodingResult(tokens[b, :n[b]].tolist(), float(avg[b]), float(nsp[b])) for b in range(B)]

    def detect_language(self) -> list[tuple[int, float]]:
        """WhisperModel.detectLanguage for the clips of the last encode(): (language index, probability)."""
        st = self.special
        if not st.is_multilingual:
            return [(0, 1.0)] * self.B
        idx = np.zeros(self.B, np.int32)
        pr = np.zeros(self.B, np.float32)
        self.ctx.check(self.ctx.lib.mia_whisper_detect_language(self.h, st.sot, st.num_languages, idx.ctypes.data, pr.ctypes.data))
        return [(int(i), float(p)) for i, p in zip(idx, pr)]

    def transcribe_windows(self, clips, o: DecodingOptions, pad_right: int = _audio.N_SAMPLES) -> list[DecodingResult]:
        """log-mel -> encode -> greedy decode of ONE 30 s window per clip (the body of WhisperSTT.transcribe's loop)."""
        clips = [np.ascontiguousarray(c, np.float32) for c in clips]
        B = len(clips)
        offs = np.zeros(B + 1, np.int64)
        np.cumsum([c.shape[0] for c in clips], out=offs[1:])
        pcm = np.concatenate(clips)
        co, keep = self._opts(o)
        tokens = np.zeros((B, o.max_tokens), np.int32)
        n = np.zeros(B, np.int32)
        avg = np.zeros(B, np.float32)
        nsp = np.zeros(B, np.float32)
        self.ctx.check(self.ctx.lib.mia_whisper_transcribe_windows(self.h, pcm.ctypes.data, offs.ctypes.data, B, pad_right, C.byref(co),
                                                                   tokens.ctypes.data, n.ctypes.data, avg.ctypes.data, nsp.ctypes.data,
                                                                   _lib.MEM_HOST))
        self.B = B
        return [DecodingResult(tokens[b, :n[b]].tolist(), float(avg[b]), float(nsp[b])) for b in range(B)]


    def transcribe_windows_device(self, pcm_ptr: int, offs: np.ndarray, o: DecodingOptions, tokens_ptr: int, n_ptr: int,
                                  avg_ptr: int, nsp_ptr: int, pad_right: int = _audio.N_SAMPLES) -> None:
        """Same as transcribe_windows but every buffer already lives in HBM (raw device pointers, e.g. torch
        tensor.data_ptr()); nothing is copied and the call only enqueues work on the ctx stream (plus the small
        host syncs of the decode loop's early-exit poll)."""
        offs = np.ascontiguousarray(offs, np.int64)
        co, keep = self._opts(o)     # three small arrays: rebuilt per call, so mutating `o` between calls takes effect
        self.B = len(offs) - 1
        self.ctx.check(self.ctx.lib.mia_whisper_transcribe_windows(self.h, pcm_ptr, offs.ctypes.data, self.B, pad_right, C.byref(co),
                                                                   tokens_ptr, n_ptr, avg_ptr, nsp_ptr, _lib.MEM_DEVICE))


    def encode_windows_device(self, pcm_ptr: int, offs: np.ndarray, pad_right: int = _audio.N_SAMPLES) -> None:
        """The first half of transcribe_windows_device (log-mel + encoder, mia_whisper_encode_windows): only enqueues."""
        offs = np.ascontiguousarray(offs, np.int64)
        self.B = len(offs) - 1
        self.ctx.check(self.ctx.lib.mia_whisper_encode_windows(self.h, pcm_ptr, offs.ctypes.data, self.B, pad_right, _lib.MEM_DEVICE))

    def decode_greedy_device(self, o: DecodingOptions, tokens_ptr: int, n_ptr: int, avg_ptr: int, nsp_ptr: int) -> None:
        """The second half: greedy decode of the batch encoded last, outputs to device pointers."""
        co, keep = self._opts(o)
        self.ctx.check(self.ctx.lib.mia_whisper_decode_greedy(self.h, C.byref(co), tokens_ptr, n_ptr, avg_ptr, nsp_ptr, _lib.MEM_DEVICE))


class GreedyDecoder:
    """GreedyDecoder(model:tokenizer:options:).decode(mel) (WhisperDecoding.swift:80-389), batched."""

    def __init__(self, model: WhisperModel, options: DecodingOptions):
        self.model, self.options = model, options

    def decode(self, mel: np.ndarray) -> list[DecodingResult]:
        self.model.encode(mel)
        return self.model.decode_greedy(self.options)
