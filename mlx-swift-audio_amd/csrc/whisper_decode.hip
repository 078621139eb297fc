// whisper_decode.hip -- host orchestration of the batched greedy decoder: state set-up, one captured hipGraph per
// decode step, early-exit polling, output compaction.  Replaces the control flow of GreedyDecoder.decode
// (STT/Whisper/WhisperDecoding.swift:96-389) for B clips at once; parity is per clip against the batch-1 semantics.
//
// Prefill note: the reference feeds the n_initial forced tokens in one causal pass; here they are fed one position
// per step through the same graph (mathematically identical: causal attention only sees earlier positions).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "decode.h"

namespace {

int pick_split(int K, int want) {  // largest S <= want with K % (32*S) == 0
  for (int s = want; s > 1; --s) if (K % (32 * s) == 0) return s;
  return 1;
}

// teacher-forced alignment pass (mia_whisper_align): cross-attention scores of the alignment heads are kept, and the decode head is
// replaced by "probability of the next given token + advance"
struct AlignHook {
  float* qk = nullptr;                    // [B][n_slots][n_ctx][T]
  const int32_t* head_slot = nullptr;     // device [L][H]: slot or -1
  int n_slots = 0;
  const int32_t* n_tok = nullptr;         // device [B]
  float* probs = nullptr;                 // device [B][n_ctx]
  int eot = 0;
};

__global__ __launch_bounds__(256) void align_token_prob(const float* __restrict__ logits, const int32_t* __restrict__ tokens, int32_t* __restrict__ pos_arr,
                                                        const int32_t* __restrict__ n_tok, float* __restrict__ probs, int V, int n_ctx, int eot) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = pos_arr[b];
  if (pos + 1 < n_tok[b]) {     // logits at `pos` predict tokens[pos + 1]: softmax over [0, eot)   (WhisperTiming.swift:600-611)
    const float* lg = logits + (int64_t)b * V;
    float m = -INFINITY;
    for (int i = tid; i < eot; i += 256) m = fmaxf(m, lg[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < eot; i += 256) sum += expf(lg[i] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    if (tid == 0) {
      const int t = tokens[(int64_t)b * n_ctx + pos + 1];
      probs[(int64_t)b * n_ctx + pos] = t < eot ? expf(lg[t] - m) / sum : 0.f;
      pos_arr[b] = pos + 1;
    }
  }
}

// enqueue one decoder step (consumes the token at st->pos, produces the token at st->pos + 1)
int enqueue_step(mia_whisper* w, const DecodeParams& p, const AlignHook* hook = nullptr) {
  hipStream_t s = w->ctx->stream;
  const int B = w->cur_B, D = p.D, H = p.H, C = p.n_ctx, T = w->dims.n_audio_ctx;
  const uint16_t* dh = (const uint16_t*)w->dh;
  // A operands (dh, da, dg) and weights (LinearW::wf) are in MFMA-fragment order (decode.h); out_frag: the output is the next GEMM's A
  auto skinny = [&](const uint16_t* A, int64_t lda, const LinearW& lw, bool use_bias, void* out, int64_t ldo, int S, int act, int mode,
                    uint16_t* ck = nullptr, uint16_t* cv = nullptr, int out_frag = 0, const float* stat_in = nullptr, const LNW* next_ln = nullptr,
                    float* stat_out = nullptr, const float* c1 = nullptr, const float* c2 = nullptr) {
    SkinnyArgs a{A, lda, (const uint16_t*)lw.wf, use_bias ? lw.b : nullptr, out, ldo, ck, cv, w->clip.pos, B, lw.N, lw.K, S, act, D, H, C};
    a.out_frag = out_frag;
    a.w_keep = w->weight_sharing;
    if (stat_in) { a.ss_in = stat_in; a.ss_tiles = D / 16; a.ss_dim = D; a.eps = 1e-5f; a.c1 = c1 ? c1 : lw.c1; a.c2 = c2 ? c2 : lw.c2; }
    if (mode == SK_RESID) { a.xres = w->dx; a.nw = next_ln->g; a.ss_out = stat_out; }
    return dec_launch_skinny(w, a, mode, s);
  };
  const int S_d = pick_split(D, 2), S_4d = pick_split(4 * D, 4);   // x 4 waves of intra-workgroup split-K each
  // LayerNorm carried across the chain (decode.h): the three residual-writing projections of a layer (self-attention out, cross-attention
  // out, fc2) add into x, store x * gamma of the NEXT LayerNorm as the next GEMM's operand plus per-tile (sum x, sum x^2); the GEMM that
  // consumes it applies mean / rstd / beta through its folded constants.  Used for the two attention output projections (8 of the 12
  // reduce + LayerNorm launches of a step go); fc2 keeps the split form (below).
  const bool fused_ln = D % 32 == 0 && w->dec[0].qkv.c1 != nullptr;
  float* st_a = w->dstat;                                          // two alternating buffers: a producer never overwrites what its
  float* st_b = w->dstat + (size_t)(D / 16) * w->cap_B * 2;        // own consumer is still reading (the chain is strictly serial anyway)
  // (the split greedy head embeds the next position itself: only the very first step needs this launch, done by the caller)
  if (hook || !dec_head_is_split(p)) dec_launch_embed_ln(w, w->dec[0].attn_ln, s);
  for (int l = 0; l < p.L; ++l) {
    const DecBlockW& b = w->dec[l];
    uint16_t* sk = (uint16_t*)w->self_k + (size_t)l * w->cap_B * C * D;
    uint16_t* sv = (uint16_t*)w->self_v + (size_t)l * w->cap_B * C * D;
    const uint16_t* xk = (const uint16_t*)w->cross_k + (size_t)l * w->cap_B * T * D;
    const uint16_t* xv = (const uint16_t*)w->cross_v + (size_t)l * w->cap_B * T * D;
    const LNW& next_ln = l + 1 < p.L ? w->dec[l + 1].attn_ln : w->dec_ln;
    if (fused_ln) {
      // self attention (its input is always a normalised row: the embedding kernel's, or the previous layer's reduce + LayerNorm)
      if (skinny(dh, D, b.qkv, true, w->dq, D, 1, MIA_ACT_NONE, SK_QKV, sk, sv)) return -1;
      if (dec_launch_attention(w, w->dq, sk, sv, w->da, 0, C, s)) return -1;
      if (skinny((const uint16_t*)w->da, D, b.out, true, w->dh, D, 1, MIA_ACT_NONE, SK_RESID, nullptr, nullptr, 0, nullptr, &b.cross_ln, st_a)) return -1;
      // cross attention (K/V primed by the encode call)
      if (skinny(dh, D, b.cq, true, w->dq, D, 1, MIA_ACT_NONE, SK_OUT16, nullptr, nullptr, 0, st_a)) return -1;
      if (dec_launch_attention(w, w->dq, xk, xv, w->da, T, T, s, hook ? hook->qk : nullptr, hook ? hook->head_slot + (size_t)l * H : nullptr,
                               hook ? hook->n_slots : 0, C)) return -1;
      if (skinny((const uint16_t*)w->da, D, b.cout, true, w->dh, D, 1, MIA_ACT_NONE, SK_RESID, nullptr, nullptr, 0, nullptr, &b.mlp_ln, st_b)) return -1;
      // MLP.  fc2 keeps its cross-workgroup split + the reduce / LayerNorm kernel: its 13 MB over the 80 workgroups an unsplit projection
      // has run at 1.1 TB/s (12 us measured against 7.5 + 4.9 split), and its consumers (the next q|k|v, the logits GEMM) then read a
      // normalised row and pay nothing.
      if (skinny(dh, D, b.mlp1, true, w->dg, 4 * D, 1, MIA_ACT_GELU, SK_OUT16, nullptr, nullptr, 1, st_b)) return -1;
      if (skinny((const uint16_t*)w->dg, 4 * D, b.mlp2, false, w->partial, 0, S_4d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
      dec_launch_reduce_ln(w, S_4d, b.mlp2.b, next_ln, s);
      continue;
    }
    // self attention
    if (skinny(dh, D, b.qkv, true, w->dq, D, 1, MIA_ACT_NONE, SK_QKV, sk, sv)) return -1;
    if (dec_launch_attention(w, w->dq, sk, sv, w->da, 0, C, s)) return -1;
    if (skinny((const uint16_t*)w->da, D, b.out, false, w->partial, 0, S_d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
    dec_launch_reduce_ln(w, S_d, b.out.b, b.cross_ln, s);
    // cross attention (K/V primed by the encode call)
    if (skinny(dh, D, b.cq, true, w->dq, D, 1, MIA_ACT_NONE, SK_OUT16)) return -1;
    if (dec_launch_attention(w, w->dq, xk, xv, w->da, T, T, s, hook ? hook->qk : nullptr, hook ? hook->head_slot + (size_t)l * H : nullptr,
                             hook ? hook->n_slots : 0, C)) return -1;
    if (skinny((const uint16_t*)w->da, D, b.cout, false, w->partial, 0, S_d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
    dec_launch_reduce_ln(w, S_d, b.cout.b, b.mlp_ln, s);
    // MLP
    if (skinny(dh, D, b.mlp1, true, w->dg, 4 * D, 1, MIA_ACT_GELU, SK_OUT16, nullptr, nullptr, 1)) return -1;
    if (skinny((const uint16_t*)w->dg, 4 * D, b.mlp2, false, w->partial, 0, S_4d, MIA_ACT_NONE, SK_PARTIAL)) return -1;
    dec_launch_reduce_ln(w, S_4d, b.mlp2.b, next_ln, s);
  }
  {  // logits = ln(x) . E^T (tied embedding, TextDecoder.swift:93)
    LinearW e; e.w = w->tok_emb; e.wf = w->tok_emb_f; e.N = p.V; e.K = D; e.c1 = w->emb_c1; e.c2 = w->emb_c2;
    if (skinny(dh, D, e, false, w->logits, p.V, 1, MIA_ACT_NONE, SK_OUTF32)) return -1;
  }
  if (p.trace && !hook && dec_launch_trace(w, s)) return -1;
  if (hook) {
    hipLaunchKernelGGL(align_token_prob, dim3(B), dim3(256), 0, s, w->logits, w->tokens, w->clip.pos, hook->n_tok, hook->probs, p.V, C, hook->eot);
    return hipGetLastError() == hipSuccess ? 0 : -1;
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
  p.trace = w->trace ? 1 : 0; p.head_single = (w->debug_flags & 2) ? 1 : 0;
  if (w->trace)
    for (int i = 0; i < w->trace_n; ++i) MIA_CHECK_ARG(ctx, w->trace_clip_ids[i] < w->cur_B, "decode: traced clip %d is not in this batch of %d", w->trace_clip_ids[i], w->cur_B);

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
  p.greedy = any_sampling ? 0 : 1;
  MIA_HIP(ctx, hipMemsetAsync(w->n_gen, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->last_ts, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->sum_logprob, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->n_logprob, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->no_speech, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(w->clip.pos, 0, (size_t)B * 4, s));

  // ---- hipGraphs per (batch, rule set): every kernel reads per-clip positions from device memory, so the same graph replays for
  // every step.  Two are kept: one step, and DEC_GRAPH_STEPS consecutive steps (between two replays the queue idles for ~8.5 us,
  // between two nodes of one graph it does not: 447 steps cost 56 + 7 replay gaps instead of 447).
  const bool no_graph = (w->debug_flags & 1) != 0;
  constexpr int DEC_GRAPH_STEPS = 8;
  if (!no_graph && (!w->graph_valid || memcmp(&w->graph_params, &p, sizeof(p)) != 0)) {
    if (w->step_graph) { (void)hipGraphExecDestroy(w->step_graph); w->step_graph = nullptr; }
    if (w->step_graph_n) { (void)hipGraphExecDestroy(w->step_graph_n); w->step_graph_n = nullptr; }
    auto capture = [&](int n_steps, hipGraphExec_t* exec) -> int {
      hipGraph_t graph = nullptr;
      MIA_HIP(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int erc = 0;
      for (int i = 0; i < n_steps && erc == 0; ++i) erc = enqueue_step(w, p);
      hipError_t ce = hipStreamEndCapture(s, &graph);
      if (erc != 0 || ce != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return mia_fail(ctx, MIA_ERR_DEVICE, "decode: step graph capture failed (%s)", hipGetErrorString(ce));
      }
      hipError_t ie = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ie != hipSuccess) return mia_fail(ctx, MIA_ERR_DEVICE, "decode: hipGraphInstantiate failed (%s)", hipGetErrorString(ie));
      return MIA_OK;
    };
    if (int rc = capture(1, &w->step_graph)) return rc;
    if (int rc = capture(DEC_GRAPH_STEPS, &w->step_graph_n)) return rc;
    w->graph_params = p;
    w->graph_valid = true;
  }

  if (dec_head_is_split(p)) dec_launch_embed_ln(w, w->dec[0].attn_ln, s);   // position 0; later positions are embedded by the head
  int prof_rec = mia_prof_begin(ctx, MIA_PROF_DECODE, 0.0);
  int steps_run = 0;
  const int min_init = *std::min_element(n_init.begin(), n_init.end());
  for (int step = 0; step < total_steps;) {
    int n = 1;
    if (no_graph) { if (enqueue_step(w, p) != 0) return mia_fail(ctx, MIA_ERR_DEVICE, "decode: step launch failed"); }
    else if (total_steps - step >= DEC_GRAPH_STEPS) { n = DEC_GRAPH_STEPS; MIA_HIP(ctx, hipGraphLaunch(w->step_graph_n, s)); }
    else MIA_HIP(ctx, hipGraphLaunch(w->step_graph, s));
    step += n;
    steps_run += n;
    // early exit: poll the finished flags every 16 steps once generation has started
    if (step > min_init && (step & 15) == 0 && step < total_steps) {
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

// ---- word-timestamp alignment -------------------------------------------------------------------------------
namespace {

// softmax over the first F_b frames of every (clip, head, token) row, in place   (WhisperTiming.swift:658)
__global__ __launch_bounds__(256) void align_softmax_rows(float* __restrict__ qk, const int32_t* __restrict__ n_tok, const int32_t* __restrict__ n_fr,
                                                          int n_slots, int n_ctx, int T) {
  __shared__ float red[8];
  const int tok = blockIdx.x, slot = blockIdx.y, b = blockIdx.z;
  if (tok >= n_tok[b]) return;
  const int F = n_fr[b];
  float* row = qk + (((int64_t)b * n_slots + slot) * n_ctx + tok) * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -INFINITY;
  for (int i = tid; i < F; i += 256) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = tid; i < F; i += 256) sum += expf(row[i] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  for (int i = tid; i < F; i += 256) row[i] = expf(row[i] - m) / sum;
}

// (w - mean_tokens) / sqrt(var_tokens + 1e-8) per (clip, head, frame) column, population variance   (:663-666)
__global__ __launch_bounds__(256) void align_standardize(float* __restrict__ qk, const int32_t* __restrict__ n_tok, const int32_t* __restrict__ n_fr,
                                                         int n_slots, int n_ctx, int T) {
  const int f = blockIdx.x * 256 + threadIdx.x, slot = blockIdx.y, b = blockIdx.z;
  if (f >= n_fr[b]) return;
  const int n = n_tok[b];
  float* col = qk + ((int64_t)b * n_slots + slot) * n_ctx * T + f;
  float s = 0.f;
  for (int t = 0; t < n; ++t) s += col[(int64_t)t * T];
  const float mean = s / (float)n;
  float q = 0.f;
  for (int t = 0; t < n; ++t) { const float d = col[(int64_t)t * T] - mean; q += d * d; }
  const float inv = 1.0f / sqrtf(q / (float)n + 1e-8f);
  for (int t = 0; t < n; ++t) col[(int64_t)t * T] = (col[(int64_t)t * T] - mean) * inv;
}

__device__ __forceinline__ void cswap(float& a, float& b) { const float lo = fminf(a, b), hi = fmaxf(a, b); a = lo; b = hi; }
__device__ __forceinline__ float median7(float v0, float v1, float v2, float v3, float v4, float v5, float v6) {
  // 7-input sorting network (16 compare-exchanges), element 3 is the median
  cswap(v0, v6); cswap(v2, v3); cswap(v4, v5);
  cswap(v0, v2); cswap(v1, v4); cswap(v3, v6);
  cswap(v0, v1); cswap(v2, v5); cswap(v3, v4);
  cswap(v1, v2); cswap(v4, v6);
  cswap(v2, v3); cswap(v4, v5);
  cswap(v1, v2); cswap(v3, v4); cswap(v5, v6);
  return v3;
}

// width-7 median along frames with reflect padding, mean over heads, negated (DTW cost)   (:683-722, medianFilterAttention :191-253)
__global__ __launch_bounds__(256) void align_median_mean(const float* __restrict__ qk, const int32_t* __restrict__ n_tok, const int32_t* __restrict__ n_fr,
                                                         int n_slots, int n_ctx, int T, float* __restrict__ cost) {
  const int f = blockIdx.x * 256 + threadIdx.x, tok = blockIdx.y, b = blockIdx.z;
  const int F = n_fr[b];
  if (f >= F || tok >= n_tok[b]) return;
  float acc = 0.f;
  for (int sl = 0; sl < n_slots; ++sl) {
    const float* row = qk + (((int64_t)b * n_slots + sl) * n_ctx + tok) * T;
    float v[7];
#pragma unroll
    for (int d = -3; d <= 3; ++d) {
      int i = f + d;
      if (i < 0) i = -i;
      if (i >= F) i = 2 * F - i - 2;
      i = i < 0 ? 0 : (i > F - 1 ? F - 1 : i);
      v[d + 3] = row[i];
    }
    acc += median7(v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
  }
  cost[((int64_t)b * n_ctx + tok) * T + f] = -(acc / (float)n_slots);
}

// dtw + backtrace (WhisperTiming.swift:46-130), host code like the reference's; cost is [N][ld] with M valid columns
void dtw_host(const float* cost, int N, int M, int64_t ld, std::vector<int32_t>& ti, std::vector<int32_t>& tj) {
  const int R = N + 1, Cc = M + 1;
  std::vector<float> acc((size_t)R * Cc, INFINITY);
  std::vector<int8_t> tr((size_t)R * Cc, -1);
  acc[0] = 0.f;
  for (int j = 1; j <= M; ++j)
    for (int i = 1; i <= N; ++i) {
      const float c0 = acc[(size_t)(i - 1) * Cc + j - 1], c1 = acc[(size_t)(i - 1) * Cc + j], c2 = acc[(size_t)i * Cc + j - 1];
      float c; int8_t t;
      if (c0 < c1 && c0 < c2) { c = c0; t = 0; } else if (c1 < c0 && c1 < c2) { c = c1; t = 1; } else { c = c2; t = 2; }
      acc[(size_t)i * Cc + j] = cost[(int64_t)(i - 1) * ld + (j - 1)] + c;
      tr[(size_t)i * Cc + j] = t;
    }
  for (int j = 0; j < Cc; ++j) tr[j] = 2;
  for (int i = 0; i < R; ++i) tr[(size_t)i * Cc] = 1;
  int i = N, j = M;
  ti.clear(); tj.clear();
  while (i > 0 || j > 0) {
    ti.push_back(i - 1); tj.push_back(j - 1);
    const int8_t t = tr[(size_t)i * Cc + j];
    if (t == 0) { --i; --j; } else if (t == 1) --i; else if (t == 2) --j; else { if (i > 0) --i; if (j > 0) --j; }
  }
  std::reverse(ti.begin(), ti.end()); std::reverse(tj.begin(), tj.end());
}

}  // namespace

extern "C" int mia_whisper_align(mia_whisper* w, const int32_t* tokens, int stride, const int32_t* n_tokens, const int32_t* heads, int n_heads,
                                 const int32_t* num_frames, int row_start, int eot, float* token_probs, int32_t* text_idx, int32_t* time_idx,
                                 int32_t* path_len, int path_cap, float* matrix) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  const mia_whisper_dims& d = w->dims;
  const int B = w->cur_B, C = d.n_text_ctx, T = d.n_audio_ctx, H = d.n_text_head, L = d.n_text_layer;
  MIA_CHECK_ARG(ctx, B > 0, "align: no audio features (call mia_whisper_encode first)");
  MIA_CHECK_ARG(ctx, tokens && n_tokens && heads && num_frames && token_probs && text_idx && time_idx && path_len, "align: null argument");
  MIA_CHECK_ARG(ctx, n_heads > 0 && n_heads <= 64 && stride > 0 && stride <= C && row_start >= 0 && eot > 0 && eot <= d.n_vocab && path_cap > 0, "align: bad sizes");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  std::vector<int32_t> slot((size_t)L * H, -1), ntok(B), nfr(B), init((size_t)B * C, 0);
  for (int i = 0; i < n_heads; ++i) {
    const int l = heads[2 * i], h = heads[2 * i + 1];
    MIA_CHECK_ARG(ctx, l >= 0 && l < L && h >= 0 && h < H, "align: alignment head (%d, %d) out of range", l, h);
    slot[(size_t)l * H + h] = i;
  }
  int max_tok = 0;
  for (int b = 0; b < B; ++b) {
    ntok[b] = n_tokens[b];
    MIA_CHECK_ARG(ctx, ntok[b] >= 2 && ntok[b] <= stride && row_start < ntok[b] - 1, "align: clip %d: token count %d out of range", b, ntok[b]);
    nfr[b] = num_frames[b] / 2;                               // stride-2 convolution (WhisperTiming.swift:622)
    MIA_CHECK_ARG(ctx, nfr[b] >= 2 && nfr[b] <= T, "align: clip %d: frame count out of range", b);
    for (int i = 0; i < ntok[b]; ++i) {
      const int32_t t = tokens[(size_t)b * stride + i];
      MIA_CHECK_ARG(ctx, t >= 0 && t < d.n_vocab, "align: token out of vocabulary");
      init[(size_t)b * C + i] = t;
    }
    max_tok = std::max(max_tok, ntok[b]);
  }
  // device scratch: qk [B][n_heads][C][T] | cost [B][C][T] | probs [B][C] | slot [L][H] | ntok [B] | nfr [B]
  const size_t n_qk = (size_t)B * n_heads * C * T, n_cost = (size_t)B * C * T;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t floats = al(n_qk) + al(n_cost) + al((size_t)B * C) + al((size_t)L * H) + 2 * al(B);
  float* ws = (float*)mia_workspace(ctx, floats * 4);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_qk = ws; float* d_cost = d_qk + al(n_qk); float* d_probs = d_cost + al(n_cost);
  int32_t* d_slot = (int32_t*)(d_probs + al((size_t)B * C)); int32_t* d_ntok = d_slot + al((size_t)L * H); int32_t* d_nfr = d_ntok + al(B);
  MIA_HIP(ctx, hipMemcpyAsync(w->tokens, init.data(), init.size() * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_slot, slot.data(), slot.size() * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_ntok, ntok.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_nfr, nfr.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemsetAsync(w->clip.pos, 0, (size_t)B * 4, s));
  MIA_HIP(ctx, hipMemsetAsync(d_probs, 0, (size_t)B * C * 4, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));       // host vectors above go out of use

  DecodeParams p{};
  p.B = B; p.V = d.n_vocab; p.D = d.n_text_state; p.H = H; p.L = L; p.n_ctx = C;
  AlignHook hook; hook.qk = d_qk; hook.head_slot = d_slot; hook.n_slots = n_heads; hook.n_tok = d_ntok; hook.probs = d_probs; hook.eot = eot;
  // teacher-forced pass, one position per step (clips that have consumed their last token idle at it)
  for (int step = 0; step < max_tok; ++step)
    if (enqueue_step(w, p, &hook) != 0) return mia_fail(ctx, MIA_ERR_DEVICE, "align: step launch failed");
  hipLaunchKernelGGL(align_softmax_rows, dim3(max_tok, n_heads, B), dim3(256), 0, s, d_qk, d_ntok, d_nfr, n_heads, C, T);
  hipLaunchKernelGGL(align_standardize, dim3((T + 255) / 256, n_heads, B), dim3(256), 0, s, d_qk, d_ntok, d_nfr, n_heads, C, T);
  hipLaunchKernelGGL(align_median_mean, dim3((T + 255) / 256, max_tok, B), dim3(256), 0, s, d_qk, d_ntok, d_nfr, n_heads, C, T, d_cost);
  MIA_HIP(ctx, hipGetLastError());
  std::vector<float> cost(n_cost);
  MIA_HIP(ctx, hipMemcpyAsync(cost.data(), d_cost, n_cost * 4, hipMemcpyDeviceToHost, s));
  std::vector<float> probs((size_t)B * C);
  MIA_HIP(ctx, hipMemcpyAsync(probs.data(), d_probs, probs.size() * 4, hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  std::vector<int32_t> ti, tj;
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < stride; ++i) token_probs[(size_t)b * stride + i] = i < C ? probs[(size_t)b * C + i] : 0.f;
    const int N = ntok[b] - 1 - row_start;       // rows [row_start, n_tok - 1): no_timestamps + text tokens, eot excluded (:724-733)
    dtw_host(cost.data() + ((size_t)b * C + row_start) * T, N, nfr[b], T, ti, tj);
    if ((int)ti.size() > path_cap) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "align: path_cap %d too small (need %zu)", path_cap, ti.size());
    path_len[b] = (int32_t)ti.size();
    for (size_t i = 0; i < ti.size(); ++i) { text_idx[(size_t)b * path_cap + i] = ti[i]; time_idx[(size_t)b * path_cap + i] = tj[i]; }
    if (matrix)
      for (int t = 0; t < ntok[b]; ++t)
        for (int f2 = 0; f2 < T; ++f2) matrix[((size_t)b * stride + t) * T + f2] = f2 < nfr[b] ? -cost[((size_t)b * C + t) * T + f2] : 0.f;
  }
  return MIA_OK;
}

// ---- test hooks ---------------------------------------------------------------------------------------------------------------------
extern "C" int mia_whisper_set_debug(mia_whisper* w, int flags) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_CHECK_ARG(w->ctx, flags >= 0 && flags <= 3, "set_debug: flags must be 0..3 (got %d)", flags);
  w->debug_flags = flags;
  return MIA_OK;
}

extern "C" int mia_whisper_trace_logits(mia_whisper* w, const int32_t* clips, int n_clips) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_CHECK_ARG(ctx, n_clips >= 0 && n_clips <= 8 && (n_clips == 0 || clips), "trace_logits: 0..8 clips");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  w->graph_valid = false;                      // the captured graphs hold the old trace pointers
  if (w->trace) { (void)hipFree(w->trace); w->trace = nullptr; }
  if (w->trace_clips) { (void)hipFree(w->trace_clips); w->trace_clips = nullptr; }
  w->trace_n = 0;
  w->trace_clip_ids.clear();
  if (n_clips == 0) return MIA_OK;
  for (int i = 0; i < n_clips; ++i) MIA_CHECK_ARG(ctx, clips[i] >= 0, "trace_logits: negative clip index");
  const size_t bytes = (size_t)n_clips * w->dims.n_text_ctx * w->dims.n_vocab * sizeof(float);
  if (hipMalloc((void**)&w->trace, bytes) != hipSuccess) { w->trace = nullptr; return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "trace_logits: hipMalloc(%zu) failed", bytes); }
  if (hipMalloc((void**)&w->trace_clips, (size_t)n_clips * 4) != hipSuccess) { (void)hipFree(w->trace); w->trace = nullptr; w->trace_clips = nullptr; return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "trace_logits: hipMalloc failed"); }
  MIA_HIP(ctx, hipMemsetAsync(w->trace, 0xff, bytes, ctx->stream));       // NaN pattern: an unwritten row cannot pass a comparison
  MIA_HIP(ctx, hipMemcpyAsync(w->trace_clips, clips, (size_t)n_clips * 4, hipMemcpyHostToDevice, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  w->trace_n = n_clips;
  w->trace_clip_ids.assign(clips, clips + n_clips);
  return MIA_OK;
}

extern "C" int mia_whisper_read_logit_trace(mia_whisper* w, int slot, int first_pos, int n_pos, float* out) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_CHECK_ARG(ctx, w->trace && slot >= 0 && slot < w->trace_n, "read_logit_trace: no trace for slot %d", slot);
  MIA_CHECK_ARG(ctx, out && first_pos >= 0 && n_pos > 0 && first_pos + n_pos <= w->dims.n_text_ctx, "read_logit_trace: bad position range");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const size_t V = w->dims.n_vocab;
  MIA_HIP(ctx, hipMemcpyAsync(out, w->trace + ((size_t)slot * w->dims.n_text_ctx + first_pos) * V, (size_t)n_pos * V * 4, hipMemcpyDeviceToHost, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
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

// The encoder half on its own stream (mia_whisper_set_encode_stream): everything between begin and end is enqueued on w->enc_stream, which
// first waits for what the context's stream holds (the previous decode still reads the cross K/V this encode rewrites); the context's
// stream then waits for the encoder's tail.  Within one handle the order of work is unchanged; across handles the encoder and decode
// queues can now carry different priorities.
struct EncStreamScope {
  mia_whisper* w; hipStream_t saved; bool on;
  explicit EncStreamScope(mia_whisper* w_) : w(w_), saved(w_->ctx->stream), on(w_->enc_stream != nullptr) {
    if (!on) return;
    (void)hipEventRecord(w->ev_enc_begin, saved);
    (void)hipStreamWaitEvent(w->enc_stream, w->ev_enc_begin, 0);
    w->ctx->stream = w->enc_stream;
  }
  ~EncStreamScope() {
    if (!on) return;
    (void)hipEventRecord(w->ev_enc_end, w->enc_stream);
    w->ctx->stream = saved;
    (void)hipStreamWaitEvent(saved, w->ev_enc_end, 0);
  }
};

extern "C" int mia_whisper_set_encode_stream(mia_whisper* w, void* hip_stream) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (w->enc_stream) MIA_HIP(ctx, hipStreamSynchronize(w->enc_stream));
  w->enc_stream = (hipStream_t)hip_stream;
  if (w->enc_stream && !w->ev_enc_begin) {
    MIA_HIP(ctx, hipEventCreateWithFlags(&w->ev_enc_begin, hipEventDisableTiming));
    MIA_HIP(ctx, hipEventCreateWithFlags(&w->ev_enc_end, hipEventDisableTiming));
  }
  return MIA_OK;
}

// Several handles on one weight copy (mia_whisper_clone) decoding at the same time: the step's weight loads then use the default cache
// policy, so that the loops that read a matrix second and third find it in the Infinity Cache; a lone decode loop streams its weights
// non-temporal (316 MB per step cycle through a 256 MB cache without a hit and only evict what the chain needs).  Measured, large-v3-turbo,
// 32 clips: three concurrent loops 4 715 -> 4 820 audio-s/s with sharing on; one loop 0.407 -> 0.423 ms per step with it on.
extern "C" int mia_whisper_set_weight_sharing(mia_whisper* w, int concurrent_readers) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  const int on = concurrent_readers > 1 ? 1 : 0;
  if (on != w->weight_sharing) { w->weight_sharing = on; w->graph_valid = false; }     // the captured step holds the launches' arguments
  return MIA_OK;
}

extern "C" int mia_whisper_encode_windows(mia_whisper* w, const float* pcm, const int64_t* offs, int B, int64_t pad_right, int mem) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_CHECK_ARG(ctx, pcm && offs && B > 0, "encode_windows: null input or B <= 0");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "encode_windows: bad mem");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  {  // a larger batch re-allocates the batch buffers: that synchronises the CONTEXT's stream, so it happens before the scope
    const int rc0 = whisper_reserve(w, B);
    if (rc0 != MIA_OK) return rc0;
  }
  EncStreamScope scope(w);
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
  return whisper_encode_from_padded_mel(w, B);
}

extern "C" int mia_whisper_transcribe_windows(mia_whisper* w, const float* pcm, const int64_t* offs, int B, int64_t pad_right,
                                              const mia_decode_opts* opts, int32_t* tokens, int32_t* n_tokens, float* avg_logprob,
                                              float* no_speech_prob, int mem) {
  const int rc = mia_whisper_encode_windows(w, pcm, offs, B, pad_right, mem);
  if (rc != MIA_OK) return rc;
  return whisper_decode(w, opts, tokens, n_tokens, avg_logprob, no_speech_prob, mem);
}
