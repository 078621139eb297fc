// whisper_decode.hip -- host orchestration of the batched greedy decoder: state set-up, one captured hipGraph per
// decode step, early-exit polling, output compaction.  Replaces the control flow of GreedyDecoder.decode
// (STT/Whisper/WhisperDecoding.swift:96-389) for B clips at once; parity is per clip against the batch-1 semantics.
//
// Prefill note: the reference feeds the n_initial forced tokens in one causal pass; here they are fed one position
// per step through the same graph (mathematically identical: causal attention only sees earlier positions).
#include <algorithm>
#include <cstdlib>

#include "decode.h"

namespace {

int pick_split(int K, int want) {  // largest S <= want with K % (32*S) == 0
  for (int s = want; s > 1; --s) if (K % (32 * s) == 0) return s;
  return 1;
}

// enqueue one decoder step (consumes the token at st->pos, produces the token at st->pos + 1)
int enqueue_step(mia_whisper* w, const DecodeParams& p) {
  hipStream_t s = w->ctx->stream;
  const int B = w->cur_B, D = p.D, H = p.H, C = p.n_ctx, T = w->dims.n_audio_ctx;
  const uint16_t* dh = (const uint16_t*)w->dh;
  auto skinny = [&](const uint16_t* A, int64_t lda, const LinearW& lw, bool use_bias, void* out, int64_t ldo, int S, int act, int mode,
                    uint16_t* ck = nullptr, uint16_t* cv = nullptr) {
    SkinnyArgs a{A, lda, (const uint16_t*)lw.w, use_bias ? lw.b : nullptr, out, ldo, ck, cv, w->clip.pos, B, lw.N, lw.K, S, act, D, H, C};
    return dec_launch_skinny(w, a, mode, s);
  };
  const int S_d = pick_split(D, 2), S_4d = pick_split(4 * D, 4);   // x 4 waves of intra-workgroup split-K each
  dec_launch_embed_ln(w, w->dec[0].attn_ln, s);
  for (int l = 0; l < p.L; ++l) {
    const DecBlockW& b = w->dec[l];
    uint16_t* sk = (uint16_t*)w->self_k + (size_t)l * w->cap_B * C * D;
    uint16_t* sv = (uint16_t*)w->self_v + (size_t)l * w->cap_B * C * D;
    const uint16_t* xk = (const uint16_t*)w->cross_k + (size_t)l * w->cap_B * T * D;
    const uint16_t* xv = (const uint16_t*)w->cross_v + (size_t)l * w->cap_B * T * D;
    // self attention
    if (skinny(dh, D, b.qkv, true, w->dq, D, 1, MIA_ACT_NONE, SK_QKV, sk, sv)) return -1;
    if (dec_launch_attention(w, w->dq, sk, sv, w->da, 0, C, s)) return -1;
    if (skinny((const uint16_t*)w->da, D, b.out, false, w->partial, 0, S_d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
    dec_launch_reduce_ln(w, S_d, b.out.b, b.cross_ln, s);
    // cross attention (K/V primed by the encode call)
    if (skinny(dh, D, b.cq, true, w->dq, D, 1, MIA_ACT_NONE, SK_OUT16)) return -1;
    if (dec_launch_attention(w, w->dq, xk, xv, w->da, T, T, s)) return -1;
    if (skinny((const uint16_t*)w->da, D, b.cout, false, w->partial, 0, S_d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
    dec_launch_reduce_ln(w, S_d, b.cout.b, b.mlp_ln, s);
    // MLP
    if (skinny(dh, D, b.mlp1, true, w->dg, 4 * D, 1, MIA_ACT_GELU, SK_OUT16)) return -1;
    if (skinny((const uint16_t*)w->dg, 4 * D, b.mlp2, false, w->partial, 0, S_4d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
    dec_launch_reduce_ln(w, S_4d, b.mlp2.b, l + 1 < p.L ? w->dec[l + 1].attn_ln : w->dec_ln, s);
  }
  {  // logits = ln(x) . E^T (tied embedding, TextDecoder.swift:93)
    LinearW e; e.w = w->tok_emb; e.N = p.V; e.K = D;
    if (skinny(dh, D, e, false, w->logits, p.V, 1, MIA_ACT_NONE, SK_OUTF32)) return -1;
  }
  if (dec_launch_head(w, w->last_ts, p, s)) return -1;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int whisper_decode(mia_whisper* w, const mia_decode_opts* o, int32_t* tokens, int32_t* n_tokens, float* avg_logprob,
                   float* no_speech_prob, int mem) {
  mia_ctx* ctx = w->ctx;
  const mia_whisper_dims& d = w->dims;
  MIA_CHECK_ARG(ctx, w->cur_B > 0, "decode: no audio features (call mia_whisper_encode first)");
  MIA_CHECK_ARG(ctx, o && o->initial_tokens && o->n_initial > 0, "decode: initial_tokens required");
  MIA_CHECK_ARG(ctx, o->max_tokens > 0 && o->max_tokens <= d.n_text_ctx, "decode: max_tokens must be in 1..n_text_ctx");
  // The Swift computes maxGenerate = maxTokens - initial and traps when that goes negative: reject instead.
  MIA_CHECK_ARG(ctx, o->n_initial < o->max_tokens, "decode: initial sequence (%d) leaves no room under max_tokens (%d)", o->n_initial, o->max_tokens);
  MIA_CHECK_ARG(ctx, o->eot >= 0 && o->eot < d.n_vocab && o->no_speech >= 0 && o->no_speech < d.n_vocab &&
                         o->timestamp_begin > 0 && o->timestamp_begin <= d.n_vocab && o->no_timestamps >= 0 && o->no_timestamps < d.n_vocab,
                "decode: special token ids out of range");
  MIA_CHECK_ARG(ctx, o->n_suppress >= 0 && o->n_blank >= 0 && (o->n_suppress == 0 || o->suppress_ids) && (o->n_blank == 0 || o->blank_ids), "decode: bad suppress tables");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "decode: bad mem");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int B = w->cur_B, C = d.n_text_ctx, V = d.n_vocab;

  DecodeParams p{};
  p.B = B; p.V = V; p.D = d.n_text_state; p.H = d.n_text_head; p.L = d.n_text_layer; p.n_ctx = C;
  p.eot = o->eot; p.no_speech = o->no_speech; p.no_timestamps = o->no_timestamps; p.timestamp_begin = o->timestamp_begin;
  p.timestamps = o->timestamps ? 1 : 0; p.max_tokens = o->max_tokens;
  p.max_initial_ts = o->max_initial_timestamp_index; p.max_new_tokens = o->max_new_tokens;

  // ---- per-clip forced prefixes, probe positions, temperatures, activity (host -> device, small)
  std::vector<int32_t> n_init(B), sot_idx(B), fin(B, 0);
  std::vector<float> temp(B);
  int total_steps = 0; bool any_sampling = false;
  {
    std::vector<int32_t> init((size_t)B * C, 0);
    for (int b = 0; b < B; ++b) {
      const int ni = (o->per_clip_initial && o->n_initial_per_clip) ? o->n_initial_per_clip[b] : o->n_initial;
      MIA_CHECK_ARG(ctx, ni > 0 && ni <= o->n_initial && ni < o->max_tokens, "decode: clip %d: bad initial length %d", b, ni);
      n_init[b] = ni;
      sot_idx[b] = o->sot_index_per_clip ? o->sot_index_per_clip[b] : o->sot_index;
      MIA_CHECK_ARG(ctx, sot_idx[b] >= 0 && sot_idx[b] < ni, "decode: clip %d: sot_index out of range", b);
      temp[b] = o->clip_temperature ? o->clip_temperature[b] : o->temperature;
      MIA_CHECK_ARG(ctx, temp[b] >= 0.0f, "decode: negative temperature");
      if (temp[b] > 0.0f) any_sampling = true;
      if (o->clip_active && !o->clip_active[b]) fin[b] = 1;
      for (int i = 0; i < ni; ++i) {
        const int32_t t = o->per_clip_initial ? o->initial_tokens[(size_t)b * o->n_initial + i] : o->initial_tokens[i];
        if (t < 0 || t >= V) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "decode: initial token %d out of vocabulary", t);
        init[(size_t)b * C + i] = t;
      }
      int cap = p.max_tokens - ni;
      if (p.max_new_tokens > 0 && p.max_new_tokens < cap) cap = p.max_new_tokens;
      if (!fin[b]) total_steps = std::max(total_steps, ni + cap - 1);
    }
    MIA_CHECK_ARG(ctx, !any_sampling || o->uniforms, "decode: temperature > 0 needs caller-provided uniforms [B][max_tokens]");
    const int nw = (V + 31) / 32;
    std::vector<uint32_t> bits((size_t)2 * nw, 0u);
    auto setbit = [&](int set, int id) { if (id >= 0 && id < V) bits[(size_t)set * nw + (id >> 5)] |= 1u << (id & 31); };
    for (int i = 0; i < o->n_suppress; ++i) { setbit(0, o->suppress_ids[i]); setbit(1, o->suppress_ids[i]); }
    for (int i = 0; i < o->n_blank; ++i) setbit(1, o->blank_ids[i]);
    setbit(1, o->eot);
    MIA_HIP(ctx, hipMemcpyAsync(w->tokens, init.data(), init.size() * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(w->suppress_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(w->clip.n_init, n_init.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(w->clip.sot_idx, sot_idx.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(w->clip.temp, temp.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(w->finished, fin.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    if (any_sampling) {
      // uniforms arrive as [B][max_tokens]; the kernel indexes [B][n_ctx]
      for (int b = 0; b < B; ++b)
        MIA_HIP(ctx, hipMemcpyAsync(w->uniforms + (size_t)b * C, o->uniforms + (size_t)b * o->max_tokens, (size_t)o->max_tokens * 4, hipMemcpyHostToDevice, s));
    }
    MIA_HIP(ctx, hipStreamSynchronize(s));   // host vectors go out of scope
  }
  MIA_HIP(ctx, hipMemsetAsync(w->n_gen, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->last_ts, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->sum_logprob, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->n_logprob, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->no_speech, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->clip.pos, 0, (size_t)B * 4, s));

  // ---- one hipGraph per (batch, rule set): every kernel reads per-clip positions from device memory
  static const bool no_graph = getenv("MIA_NO_GRAPH") != nullptr;
  if (!no_graph && (!w->graph_valid || memcmp(&w->graph_params, &p, sizeof(p)) != 0)) {
    if (w->step_graph) { (void)hipGraphExecDestroy(w->step_graph); w->step_graph = nullptr; }
    hipGraph_t graph = nullptr;
    MIA_HIP(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int erc = enqueue_step(w, p);
    hipError_t ce = hipStreamEndCapture(s, &graph);
    if (erc != 0 || ce != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      return mia_fail(ctx, MIA_ERR_DEVICE, "decode: step graph capture failed (%s)", hipGetErrorString(ce));
    }
    hipError_t ie = hipGraphInstantiate(&w->step_graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) return mia_fail(ctx, MIA_ERR_DEVICE, "decode: hipGraphInstantiate failed (%s)", hipGetErrorString(ie));
    w->graph_params = p;
    w->graph_valid = true;
  }

  int prof_rec = mia_prof_begin(ctx, MIA_PROF_DECODE, 0.0);
  int steps_run = 0;
  const int min_init = *std::min_element(n_init.begin(), n_init.end());
  for (int step = 0; step < total_steps; ++step) {
    ++steps_run;
    if (no_graph) { if (enqueue_step(w, p) != 0) return mia_fail(ctx, MIA_ERR_DEVICE, "decode: step launch failed"); }
    else MIA_HIP(ctx, hipGraphLaunch(w->step_graph, s));
    // early exit: poll the finished flags every 16 steps once generation has started
    if (step >= min_init && (step & 15) == 15 && step + 1 < total_steps) {
      MIA_HIP(ctx, hipMemcpyAsync(fin.data(), w->finished, (size_t)B * 4, hipMemcpyDeviceToHost, s));
      MIA_HIP(ctx, hipStreamSynchronize(s));
      bool all = true;
      for (int b = 0; b < B; ++b) all = all && fin[b];
      if (all) break;
    }
  }
  if (prof_rec >= 0) { ctx->prof[prof_rec].work = steps_run; mia_prof_end(ctx, prof_rec); }
  dec_launch_finalize(w, w->out_n, p, s);
  MIA_HIP(ctx, hipGetLastError());
  const hipMemcpyKind kind = mem == MIA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (tokens) MIA_HIP(ctx, hipMemcpyAsync(tokens, w->out_tokens, (size_t)B * p.max_tokens * 4, kind, s));
  if (n_tokens) MIA_HIP(ctx, hipMemcpyAsync(n_tokens, w->out_n, (size_t)B * 4, kind, s));
  if (avg_logprob) MIA_HIP(ctx, hipMemcpyAsync(avg_logprob, w->out_avg, (size_t)B * 4, kind, s));
  if (no_speech_prob) MIA_HIP(ctx, hipMemcpyAsync(no_speech_prob, w->no_speech, (size_t)B * 4, kind, s));
  if (mem == MIA_MEM_HOST) MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}

extern "C" int mia_whisper_decode_greedy(mia_whisper* w, const mia_decode_opts* opts, int32_t* tokens, int32_t* n_tokens,
                                         float* avg_logprob, float* no_speech_prob, int mem) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  return whisper_decode(w, opts, tokens, n_tokens, avg_logprob, no_speech_prob, mem);
}

// softmax / argmax over the language-token slice of the logits after one step on [sot]  (WhisperModel.swift:223-260)
static __global__ void lang_kernel(const float* __restrict__ logits, int V, int start, int n, int32_t* __restrict__ idx, float* __restrict__ prob) {
  const int b = blockIdx.x;
  const float* lg = logits + (int64_t)b * V + start;
  if (threadIdx.x == 0) {
    float m = -INFINITY; int mi = 0;
    for (int i = 0; i < n; ++i) if (lg[i] > m) { m = lg[i]; mi = i; }
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += __expf(lg[i] - m);
    idx[b] = mi;
    prob[b] = 1.0f / s;
  }
}

extern "C" int mia_whisper_detect_language(mia_whisper* w, int32_t sot, int32_t n_languages, int32_t* lang_idx, float* prob) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  const mia_whisper_dims& d = w->dims;
  MIA_CHECK_ARG(ctx, w->cur_B > 0, "detect_language: no audio features (call mia_whisper_encode first)");
  MIA_CHECK_ARG(ctx, lang_idx && prob, "detect_language: null outputs");
  MIA_CHECK_ARG(ctx, sot >= 0 && n_languages > 0 && sot + 1 + n_languages <= d.n_vocab, "detect_language: token range out of vocabulary");
  // one decoder step on [sot]; the head runs too (its outputs are ignored)
  int32_t init = sot;
  mia_decode_opts o{};
  o.initial_tokens = &init; o.n_initial = 1; o.sot_index = 0;
  o.eot = sot > 0 ? sot - 1 : 0; o.no_speech = 0; o.no_timestamps = 0; o.timestamp_begin = d.n_vocab; o.timestamps = 0;
  o.max_tokens = 2; o.max_initial_timestamp_index = 50;
  int rc = whisper_decode(w, &o, nullptr, nullptr, nullptr, nullptr, MIA_MEM_DEVICE);
  if (rc != MIA_OK) return rc;
  const int B = w->cur_B;
  hipLaunchKernelGGL(lang_kernel, dim3(B), dim3(64), 0, ctx->stream, w->logits, d.n_vocab, sot + 1, n_languages, w->out_n, w->out_avg);
  MIA_HIP(ctx, hipMemcpyAsync(lang_idx, w->out_n, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  MIA_HIP(ctx, hipMemcpyAsync(prob, w->out_avg, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIA_OK;
}

extern "C" int mia_whisper_transcribe_windows(mia_whisper* w, const float* pcm, const int64_t* offs, int B, int64_t pad_right,
                                              const mia_decode_opts* opts, int32_t* tokens, int32_t* n_tokens, float* avg_logprob,
                                              float* no_speech_prob, int mem) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_CHECK_ARG(ctx, pcm && offs && B > 0, "transcribe_windows: null input or B <= 0");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "transcribe_windows: bad mem");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = whisper_reserve(w, B);
  if (rc != MIA_OK) return rc;
  const mia_whisper_dims& d = w->dims;
  const int64_t rows = 2 * (int64_t)d.n_audio_ctx;
  const size_t sc = mia_logmel_scratch_bytes(B, rows, d.n_mels);
  const size_t pcm_bytes = (size_t)offs[B] * sizeof(float);
  char* ws = (char*)mia_workspace(ctx, sc + (mem == MIA_MEM_HOST ? align_up(pcm_bytes, 256) : 0));
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  const float* d_pcm = pcm;
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(ws + sc, pcm, pcm_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_pcm = (const float*)(ws + sc);
  }
  // log-mel straight into the zero-row-padded layout conv1 reads (row offset 1), in the compute dtype
  rc = mia_logmel_device(ctx, d_pcm, offs, B, d.n_mels, 0, pad_right, rows, w->mel_pad, w->dtype, false,
                         (rows + 2) * d.n_mels, d.n_mels, 1, 1, ws);
  if (rc != MIA_OK) return rc;
  rc = whisper_encode_from_padded_mel(w, B);
  if (rc != MIA_OK) return rc;
  return whisper_decode(w, opts, tokens, n_tokens, avg_logprob, no_speech_prob, mem);
}
