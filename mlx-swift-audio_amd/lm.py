"""Host-side mirror of the reference's causal-LM interface (OrpheusLMHeadModel / Qwen2 blocks + sampleNextToken + parseOutput),
backed by the gfx950 HIP layer (include/mia.h).  parse_output is integer bookkeeping and stays on the host, as in the Swift."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .whisper import _TensorView

END_TOKEN = 128258                    # OrpheusTTS.swift:75-85
CODE_OFFSET = 128266
AUDIO_CODE_DATA_START_MARKER = 128257
MAX_TOKEN_COUNT = 1200
REPETITION_CONTEXT_SIZE = 20


class _LmCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "hidden", "inter", "n_layers", "n_heads", "n_kv_heads", "head_dim", "max_ctx")] + \
               [("rms_eps", C.c_float), ("rope_theta", C.c_float), ("rope_llama3", C.c_int32), ("rope_factor", C.c_float),
                ("rope_low", C.c_float), ("rope_high", C.c_float), ("rope_old_ctx", C.c_int32), ("qkv_bias", C.c_int32),
                ("tie_embeddings", C.c_int32)]


class _Sampler(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("rep_penalty", C.c_float), ("rep_window", C.c_int32),
                ("max_new_tokens", C.c_int32), ("n_stop", C.c_int32), ("stop_ids", C.c_int32 * 4), ("reserved", C.c_int32)]


def _declare(lib):
    if getattr(lib, "_lm_declared", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.mia_lm_load.restype = vp
    lib.mia_lm_load.argtypes = [vp, C.POINTER(_LmCfg), C.POINTER(_TensorView), i32, i32]
    lib.mia_lm_free.restype = None
    lib.mia_lm_free.argtypes = [vp]
    lib.mia_lm_reset.restype = i32
    lib.mia_lm_reset.argtypes = [vp]
    lib.mia_lm_forward.restype = i32
    lib.mia_lm_forward.argtypes = [vp, vp, i32, vp]
    lib.mia_lm_generate.restype = i32
    lib.mia_lm_generate.argtypes = [vp, vp, i32, C.POINTER(_Sampler), vp, vp, vp]
    lib.mia_sample_top_p.restype = i32
    lib.mia_sample_top_p.argtypes = [vp, vp, i32, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    lib._lm_declared = True


class CausalLM:
    def __init__(self, ctx, h, cfg):
        self.ctx, self.h, self.cfg = ctx, h, cfg
        ctx.adopt(self)

    @staticmethod
    def load(ctx: _lib.Context, cfg, weights: dict[str, np.ndarray], dtype: int = _lib.BF16) -> "CausalLM":
        _declare(ctx.lib)
        c = _LmCfg(cfg.vocab, cfg.hidden, cfg.inter, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.max_ctx, cfg.rms_eps,
                   cfg.rope_theta, 1 if cfg.rope_llama3 else 0, cfg.rope_factor, cfg.rope_low, cfg.rope_high, cfg.rope_old_ctx,
                   1 if cfg.qkv_bias else 0, 1 if cfg.tie_embeddings else 0)
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
        h = ctx.lib.mia_lm_load(ctx.h, C.byref(c), views, len(weights), dtype)
        if not h:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, ctx.lib.mia_last_error(ctx.h).decode())
        return CausalLM(ctx, h, cfg)

    def attach_q4(self, packed: dict[str, np.ndarray], group_size: int = 64, bits: int = 4) -> None:
        """mia_lm_attach_quantized (bits 4 | 8): `packed` holds every step Linear as the checkpoint stores it -- `<name>.weight` uint32 codes,
        `<name>.scales` / `<name>.biases` float16 (or uint16 arrays tagged .st_dtype == "BF16", as checkpoint.read_safetensors returns
        bf16 payloads).  The handle must have been loaded from the de-quantised tensors of the same checkpoint."""
        lib = self.ctx.lib
        lib.mia_lm_attach_quantized.restype = C.c_int
        lib.mia_lm_attach_quantized.argtypes = [C.c_void_p, C.POINTER(_TensorView), C.c_int, C.c_int, C.c_int]
        views = (_TensorView * len(packed))()
        keep = []
        for i, (name, arr) in enumerate(packed.items()):
            a = np.ascontiguousarray(arr)
            if a.dtype == np.uint32:
                dt = _lib.U32
            elif a.dtype == np.float16:
                dt = _lib.F16
            elif a.dtype == np.uint16 and getattr(arr, "st_dtype", "BF16") == "BF16":
                dt = _lib.BF16
            else:
                raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, f"attach_q4: '{name}' must be uint32 codes or 16-bit scales / biases (got {a.dtype})")
            keep.append(a)
            shp = (C.c_int64 * 4)(*(list(a.shape) + [0] * (4 - a.ndim)))
            views[i] = _TensorView(name.encode(), dt, a.ndim, shp, a.ctypes.data)
        self.ctx.check(lib.mia_lm_attach_quantized(self.h, views, len(packed), group_size, bits))

    def set_debug(self, flags: int) -> None:
        """Test hook (mia_lm_set_debug): bit 0 = no hipGraph, bit 1 = prompts token by token (no batched prompt pass)."""
        lib = self.ctx.lib
        lib.mia_lm_set_debug.restype = C.c_int
        lib.mia_lm_set_debug.argtypes = [C.c_void_p, C.c_int]
        self.ctx.check(lib.mia_lm_set_debug(self.h, int(flags)))

    def use_q4(self, on: bool) -> None:
        lib = self.ctx.lib
        lib.mia_lm_use_q4.restype = C.c_int
        lib.mia_lm_use_q4.argtypes = [C.c_void_p, C.c_int]
        self.ctx.check(lib.mia_lm_use_q4(self.h, 1 if on else 0))

    def close(self):
        if self.h and getattr(self.ctx, 'h', None):
            self.ctx.lib.mia_lm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.ctx.check(self.ctx.lib.mia_lm_reset(self.h))

    def forward(self, ids) -> np.ndarray:
        """model(ids, cache)[0, -1]: fp32 logits after the last of `ids` (appended to the KV cache)."""
        a = np.ascontiguousarray(ids, np.int32)
        out = np.empty(self.cfg.vocab, np.float32)
        self.ctx.check(self.ctx.lib.mia_lm_forward(self.h, a.ctypes.data, a.size, out.ctypes.data))
        return out

    def generate(self, prompt, uniforms, temperature=0.6, top_p=0.8, rep_penalty=1.3, rep_window=REPETITION_CONTEXT_SIZE,
                 max_new_tokens=MAX_TOKEN_COUNT, stop_ids=(END_TOKEN,)) -> list[int]:
        a = np.ascontiguousarray(prompt, np.int32)
        u = np.ascontiguousarray(uniforms, np.float32)
        if u.size < max_new_tokens:
            raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "need one uniform per possible draw")
        sp = _Sampler(temperature, top_p, rep_penalty, rep_window, max_new_tokens, len(stop_ids), (C.c_int32 * 4)(*(list(stop_ids) + [0] * (4 - len(stop_ids)))), 0)
        out = np.zeros(max_new_tokens, np.int32)
        n = C.c_int32(0)
        self.ctx.check(self.ctx.lib.mia_lm_generate(self.h, a.ctypes.data, a.size, C.byref(sp), u.ctypes.data, out.ctypes.data, C.byref(n)))
        return out[:n.value].tolist()


def _declare_batch(lib):
    if not getattr(lib, "_lm_batch_declared", False):
        lib.mia_lm_set_batch.restype = C.c_int
        lib.mia_lm_set_batch.argtypes = [C.c_void_p, C.c_int]
        lib.mia_lm_generate_batch.restype = C.c_int
        lib.mia_lm_generate_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Sampler), C.c_void_p, C.c_void_p, C.c_void_p]
        lib._lm_batch_declared = True


def _set_batch(self, max_batch: int) -> None:
    """Size the per-sequence state for up to `max_batch` (<= 32) sequences decoded side by side."""
    _declare_batch(self.ctx.lib)
    self.ctx.check(self.ctx.lib.mia_lm_set_batch(self.h, int(max_batch)))


def _generate_batch(self, prompts, uniforms, temperature=0.6, top_p=0.8, rep_penalty=1.3, rep_window=REPETITION_CONTEXT_SIZE,
                    max_new_tokens=MAX_TOKEN_COUNT, stop_ids=(END_TOKEN,)) -> list[list[int]]:
    """generate() for several prompts at once (sentence-level batching): uniforms [n_seq, max_new_tokens]; sequence b's ids equal
    generate(prompts[b], uniforms[b]).  Call set_batch(n) first."""
    u = np.ascontiguousarray(uniforms, np.float32)
    n_seq = len(prompts)
    if u.ndim != 2 or u.shape[0] != n_seq or u.shape[1] < max_new_tokens:
        raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "uniforms must be [n_seq, >= max_new_tokens]")
    u = np.ascontiguousarray(u[:, :max_new_tokens])
    offs = np.zeros(n_seq + 1, np.int32)
    np.cumsum([len(p) for p in prompts], out=offs[1:])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(p, np.int32) for p in prompts]))
    sp = _Sampler(temperature, top_p, rep_penalty, rep_window, max_new_tokens, len(stop_ids), (C.c_int32 * 4)(*(list(stop_ids) + [0] * (4 - len(stop_ids)))), 0)
    out = np.zeros((n_seq, max_new_tokens), np.int32)
    n = np.zeros(n_seq, np.int32)
    _declare_batch(self.ctx.lib)
    self.ctx.check(self.ctx.lib.mia_lm_generate_batch(self.h, flat.ctypes.data, offs.ctypes.data, n_seq, C.byref(sp), u.ctypes.data, out.ctypes.data, n.ctypes.data))
    return [out[b, :n[b]].tolist() for b in range(n_seq)]


CausalLM.set_batch = _set_batch
CausalLM.generate_batch = _generate_batch


def _generate_ras(self, prompt_embeds, uniforms, min_len, max_len, eos, top_p=0.8, top_k=25, win=10, tau=0.1) -> list[int]:
    """mia_lm_generate_ras: embedding-row prompt -> RAS-sampled ids (stops at `eos` once min_len ids are out, or at max_len)."""
    lib = self.ctx.lib
    if not getattr(lib, "_ras_declared", False):
        lib.mia_lm_generate_ras.restype = C.c_int
        lib.mia_lm_generate_ras.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Ras), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib._ras_declared = True
    x = np.ascontiguousarray(prompt_embeds, np.float32)
    rp = _Ras(top_p, top_k, win, tau, eos, min_len, max_len)
    u = np.ascontiguousarray(uniforms, np.float32)
    out = np.zeros(max_len + 1, np.int32)
    n = C.c_int32(0)
    self.ctx.check(lib.mia_lm_generate_ras(self.h, x.ctypes.data, x.shape[0], C.byref(rp), u.ctypes.data, u.size, out.ctypes.data, C.byref(n)))
    return out[:n.value].tolist()


CausalLM.generate_ras = _generate_ras


def _generate_ras_batch(self, prompt_embeds, uniforms, min_lens, max_lens, eos, top_p=0.8, top_k=25, win=10, tau=0.1) -> list[list[int]]:
    """mia_lm_generate_ras_batch: one embedding-row prompt, (min_len, max_len) pair and uniform row per utterance; set_batch(n) first."""
    lib = self.ctx.lib
    if not getattr(lib, "_ras_batch_declared", False):
        lib.mia_lm_generate_ras_batch.restype = C.c_int
        lib.mia_lm_generate_ras_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(_Ras), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib._ras_batch_declared = True
    n_seq = len(prompt_embeds)
    xs = [np.ascontiguousarray(x, np.float32) for x in prompt_embeds]
    offs = np.zeros(n_seq + 1, np.int32)
    np.cumsum([x.shape[0] for x in xs], out=offs[1:])
    flat = np.ascontiguousarray(np.concatenate(xs, axis=0))
    rps = (_Ras * n_seq)(*[_Ras(top_p, top_k, win, tau, eos, int(min_lens[b]), int(max_lens[b])) for b in range(n_seq)])
    u = np.ascontiguousarray(uniforms, np.float32)
    if u.ndim != 2 or u.shape[0] != n_seq:
        raise _lib.MiaError(_lib.ERR_INVALID_ARGUMENT, "uniforms must be [n_seq, n_uniforms]")
    stride = int(max(max_lens)) + 1
    out = np.zeros((n_seq, stride), np.int32)
    n = np.zeros(n_seq, np.int32)
    self.ctx.check(lib.mia_lm_generate_ras_batch(self.h, flat.ctypes.data, offs.ctypes.data, n_seq, rps, u.ctypes.data, u.shape[1], out.ctypes.data, stride, n.ctypes.data))
    return [out[b, :n[b]].tolist() for b in range(n_seq)]


CausalLM.generate_ras_batch = _generate_ras_batch


def sample_next_token(ctx: _lib.Context, logits: np.ndarray, history, uniform: float, temperature=0.6, top_p=0.8, rep_penalty=1.3) -> int:
    """sampleNextToken(logits:history:temperature:topP:repetitionPenalty:) with an explicit uniform for the categorical draw."""
    _declare(ctx.lib)
    lg = np.ascontiguousarray(logits, np.float32)
    h = np.ascontiguousarray(history, np.int32)
    out = C.c_int32(0)
    ctx.check(ctx.lib.mia_sample_top_p(ctx.h, lg.ctypes.data, lg.size, h.ctypes.data if h.size else None, h.size, rep_penalty, temperature, top_p,
                                       uniform, C.byref(out)))
    return out.value


def parse_output(tokens: list[int]) -> list[list[int]]:
    """parseOutput (OrpheusTTS.swift:472-508): generated ids -> SNAC code lists [N, 2N, 4N] (host integer logic, as in the Swift)."""
    last = max((i for i, t in enumerate(tokens) if t == AUDIO_CODE_DATA_START_MARKER), default=-1)
    rel = tokens[last + 1:] if last >= 0 else tokens
    f = [t for t in rel if t != END_TOKEN and t >= CODE_OFFSET]
    f = [t - CODE_OFFSET for t in f[:(len(f) // 7) * 7]]
    l1, l2, l3 = [], [], []
    for i in range(len(f) // 7):
        b = 7 * i
        l1.append(f[b]); l2.append(f[b + 1] - 4096); l3.append(f[b + 2] - 2 * 4096); l3.append(f[b + 3] - 3 * 4096)
        l2.append(f[b + 4] - 4 * 4096); l3.append(f[b + 5] - 5 * 4096); l3.append(f[b + 6] - 6 * 4096)
    return [l1, l2, l3]


class _Ras(C.Structure):
    _fields_ = [("top_p", C.c_float), ("top_k", C.c_int32), ("win", C.c_int32), ("tau", C.c_float), ("eos", C.c_int32),
                ("min_len", C.c_int32), ("max_len", C.c_int32)]


class Qwen2LM:
    """Qwen2LM.inference (TTS/CosyVoice2/LLM/Qwen2LM.swift:335-376): the prompt is assembled from three embedding tables on the
    host (row gathers, no arithmetic) and handed to the device loop as embedding rows."""

    SOS_EOS, TASK_ID = 0, 1

    def __init__(self, lm: CausalLM, weights: dict[str, np.ndarray], speech_token_size: int = 6561):
        self.lm, self.speech_token_size = lm, speech_token_size
        self.text_emb = weights["model.embed_tokens.weight"]
        self.llm_emb = weights["llm_embedding.weight"]
        self.speech_emb = weights["speech_embedding.weight"]

    def lm_input(self, text, prompt_text, prompt_speech_tokens) -> np.ndarray:
        from .synthetic import round_array
        rows = [self.llm_emb[self.SOS_EOS][None], self.text_emb[np.asarray(list(prompt_text) + list(text), np.int64)],
                self.llm_emb[self.TASK_ID][None]]
        if len(prompt_speech_tokens):
            rows.append(self.speech_emb[np.asarray(prompt_speech_tokens, np.int64)])
        return np.ascontiguousarray(np.concatenate(rows, axis=0), np.float32)

    def inference(self, text, prompt_text, prompt_speech_tokens, uniforms, max_token_text_ratio=20.0, min_token_text_ratio=2.0,
                  top_p=0.8, top_k=25, win=10, tau=0.1) -> list[int]:
        x = self.lm_input(text, prompt_text, prompt_speech_tokens)
        min_len, max_len = int(len(text) * min_token_text_ratio), int(len(text) * max_token_text_ratio)
        return self.lm.generate_ras(x, uniforms, min_len, max_len, self.speech_token_size, top_p, top_k, win, tau)


class OrpheusTTS:
    """generateChunk (TTS/Orpheus/TTSEngine/OrpheusTTS.swift:224-373) from token ids onward: LM sampling loop ->
    parseOutput -> SNAC decode.  Text tokenisation / voice prefix / sentence splitting stay with the caller (CPU text code)."""

    def __init__(self, lm: CausalLM, snac):
        self.lm, self.snac = lm, snac

    def generate_chunk(self, input_ids, uniforms, noise=None, temperature=0.6, top_p=0.8, max_new_tokens=MAX_TOKEN_COUNT):
        gen = self.lm.generate(input_ids, uniforms, temperature=temperature, top_p=top_p, rep_penalty=1.3,
                               rep_window=REPETITION_CONTEXT_SIZE, max_new_tokens=max_new_tokens, stop_ids=(END_TOKEN,))
        codes = parse_output(list(input_ids) + gen)
        if not codes[0]:
            return gen, np.zeros(0, np.float32)
        n = len(codes[0])
        lim = self.snac.cfg.codebook_size
        codes = [[min(max(c, 0), lim - 1) for c in lv] for lv in codes]     # random-init LMs emit out-of-layer ids; real ones do not
        nz = noise
        if noise is not None:
            nz = np.ascontiguousarray(noise[:self.snac.noise_len(4 * n)], np.float32)
        return gen, self.snac.decode(codes, nz)

    def generate_chunks(self, input_ids_list, uniforms, noises=None, temperature=0.6, top_p=0.8, max_new_tokens=MAX_TOKEN_COUNT):
        """The sentence loop of OrpheusTTS.generate (OrpheusTTS.swift:179-191) with the sentences' LM loops side by side
        (CausalLM.generate_batch; call lm.set_batch(n) first): one (ids, pcm) pair per sentence, identical to generate_chunk per
        sentence with the same uniforms / noise.  uniforms [n, max_new_tokens]; noises: one array per sentence or None."""
        gens = self.lm.generate_batch(input_ids_list, uniforms, temperature=temperature, top_p=top_p, rep_penalty=1.3,
                                      rep_window=REPETITION_CONTEXT_SIZE, max_new_tokens=max_new_tokens, stop_ids=(END_TOKEN,))
        out = []
        lim = self.snac.cfg.codebook_size
        for b, gen in enumerate(gens):
            codes = parse_output(list(input_ids_list[b]) + gen)
            if not codes[0]:
                out.append((gen, np.zeros(0, np.float32)))
                continue
            n = len(codes[0])
            codes = [[min(max(c, 0), lim - 1) for c in lv] for lv in codes]
            nz = None if noises is None or noises[b] is None else np.ascontiguousarray(noises[b][:self.snac.noise_len(4 * n)], np.float32)
            out.append((gen, self.snac.decode(codes, nz)))
        return out
