// lm.hip -- autoregressive Llama-3 / Qwen2 style language-model decode on gfx950 (§8 a10-a12, a18).
//
// Replaces, per token, OrpheusModel / OrpheusLMHeadModel (TTS/Orpheus/BuildingBlocks/TransformerBlock.swift:70-105,129-139,
// 165-180,223-233), Llama3RoPE (TTS/Shared/Llama3RoPE.swift:27-66,104-114), SwiGLUMLP (TTS/Shared/SwiGLUMLP.swift:27-29),
// Qwen2Attention / Qwen2 blocks (TTS/CosyVoice2/LLM/Qwen2LM.swift:48-151) and the Orpheus sampler
// (TTS/Orpheus/TTSEngine/OrpheusTTS.swift:375-470): repetition penalty -> temperature -> top-p -> categorical.
//
// Decode at batch 1 is HBM-bound on the weights (Orpheus-3B: 6.6 GB bf16 per token).  Every projection is the skinny
// MFMA GEMM of decode_kernels.hip (weights HBM -> VGPR once, split-K partials summed in a fixed order by the consumer
// kernel); RMSNorm / RoPE / KV-cache write / GQA attention / SwiGLU / sampling are fused around it; one hipGraph per token.
// The prompt is consumed one position per step through the same graph (identical maths to a causal prefill).
// Stochastic stage: the categorical draw takes an explicit uniform per step (inverse CDF over the kept tokens in index
// order); the reference draws from MLX's unseeded RNG, so parity is on the kept set / distribution, not on the stream.
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "decode.h"

struct LmState { int pos; int n_hist; int finished; int n_gen; int n_embeds; int n_out; int u_cursor; int pad; };

struct LmLayer {
  float* in_norm = nullptr; float* post_norm = nullptr;
  void* wqkv = nullptr; float* bqkv = nullptr;   // [(Hq+2Hkv)*dh][hidden]
  void* wo = nullptr;                            // [hidden][Hq*dh]
  void* wgu = nullptr;                           // [2*inter][hidden], rows interleaved gate/up
  void* wdown = nullptr;                         // [hidden][inter]
};

struct mia_lm {
  mia_ctx* ctx = nullptr;
  mia_lm_config cfg{};
  int dtype = MIA_BF16;
  std::vector<void*> allocs;
  void* embed = nullptr;        // 16-bit [V][hidden]
  void* lm_head = nullptr;      // 16-bit [V][hidden] (== embed when tied)
  float* head_bias = nullptr;   // optional (CosyVoice2 llm_decoder)
  int head_vocab = 0;           // rows of lm_head (CosyVoice2: speech vocabulary + 3)
  void* gen_embed = nullptr;    // 16-bit [rows][hidden]: embedding of GENERATED ids when it differs from embed (speech_embedding)
  int gen_rows = 0;
  float* embeds = nullptr;      // fp32 [max_ctx][hidden]: caller-provided prompt embeddings (Qwen2LM.inference builds its prompt from three tables)
  int32_t* out_tokens = nullptr;  // [max_ctx] emitted tokens of the RAS loop
  float* final_norm = nullptr;
  float* inv_freq = nullptr;    // [dh/2]
  std::vector<LmLayer> layers;
  // state
  void* k_cache = nullptr; void* v_cache = nullptr;   // [L][Hkv][max_ctx][dh]
  float* x = nullptr; void* h = nullptr; float* qkv_part = nullptr; void* q = nullptr; void* att = nullptr; void* act = nullptr;
  float* partial = nullptr; float* logits = nullptr;
  int32_t* tokens = nullptr;    // [max_ctx] full sequence
  int32_t* hist = nullptr;      // [64] repetition window (ring, oldest first)
  float* uniforms = nullptr;    // [max_ctx]
  uint32_t* hist_bins = nullptr;  // sampler scratch
  LmState* state = nullptr;
  hipGraphExec_t graph = nullptr;
  mia_lm_sampler graph_sampler{};
  bool graph_valid = false;
  bool graph_sampling = false;
  int S_qkv = 1, S_o = 1, S_down = 1;
};

namespace {

constexpr int LM_NV = 4;   // float4 per thread of the 256-thread row kernels: hidden <= 4096

__device__ __forceinline__ float blk256_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename T>
__device__ __forceinline__ void rms_store(const f32x4 (&v)[LM_NV], int nv, int D, float eps, const float* __restrict__ w, uint16_t* __restrict__ h, float* sh) {
  const int tid = threadIdx.x;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) if (tid + 256 * i < nv) q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  const float rstd = rsqrtf(blk256_sum(q, sh) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) {
    const int c = tid + 256 * i;
    if (c >= nv) continue;
    const f32x4 g = *reinterpret_cast<const f32x4*>(w + 4 * c);
    *reinterpret_cast<u32x2*>(h + 4 * c) = (u32x2){pack2<T>(v[i][0] * rstd * g[0], v[i][1] * rstd * g[1]), pack2<T>(v[i][2] * rstd * g[2], v[i][3] * rstd * g[3])};
  }
}

// x = E[token[pos]] (or a caller-provided embedding row);  h = RMSNorm(x) * w
template <typename T>
__global__ __launch_bounds__(256) void lm_embed_norm(const int32_t* __restrict__ tokens, const uint16_t* __restrict__ emb, const uint16_t* __restrict__ gen_emb,
                                                     const float* __restrict__ embeds, const float* __restrict__ w,
                                                     float* __restrict__ x, uint16_t* __restrict__ h, const LmState* __restrict__ st, int D, float eps) {
  __shared__ float sh[4];
  const int tid = threadIdx.x, nv = D >> 2;
  const int pos = st->pos;
  const bool from_rows = pos < st->n_embeds;               // prompt given as embedding rows
  const int tok = from_rows ? 0 : tokens[pos];
  const uint16_t* e = ((st->n_embeds > 0 && gen_emb) ? gen_emb : emb) + (int64_t)tok * D;
  const float* er = embeds + (int64_t)pos * D;
  f32x4 v[LM_NV];
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      if (from_rows) v[i] = *reinterpret_cast<const f32x4*>(er + 4 * c);
      else {
        const s16x4 ev = *reinterpret_cast<const s16x4*>(e + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = T::to_f32((uint16_t)ev[j]);
      }
      *reinterpret_cast<f32x4*>(x + 4 * c) = v[i];
    } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  rms_store<T>(v, nv, D, eps, w, h, sh);
}

// x += sum_s partial[s];  h = RMSNorm(x) * w
template <typename T>
__global__ __launch_bounds__(256) void lm_reduce_norm(const float* __restrict__ partial, int S, const float* __restrict__ w, float* __restrict__ x,
                                                      uint16_t* __restrict__ h, int D, float eps) {
  __shared__ float sh[4];
  const int tid = threadIdx.x, nv = D >> 2;
  f32x4 v[LM_NV];
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * c);
      for (int k = 0; k < S; ++k) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(partial + (int64_t)k * D + 4 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += p[j];
      }
      v[i] = a;
      *reinterpret_cast<f32x4*>(x + 4 * c) = a;
    } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  rms_store<T>(v, nv, D, eps, w, h, sh);
}

// q|k|v = sum_s partial[s] + bias; RoPE (split-half pairs (i, i+dh/2), angle = pos * inv_freq[i]) on q and k; q -> qout,
// k, v -> cache[kv head][pos][:]          one thread per rotation pair / per v element
template <typename T>
__global__ __launch_bounds__(256) void lm_rope_cache(const float* __restrict__ part, int S, const float* __restrict__ bias, const float* __restrict__ inv_freq,
                                                     uint16_t* __restrict__ qout, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                     const LmState* __restrict__ st, int Hq, int Hkv, int dh, int max_ctx) {
  const int Nq = Hq * dh, Nk = Hkv * dh, N = Nq + 2 * Nk;
  const int half = dh >> 1;
  const int pos = st->pos;
  const int n_pairs = (Hq + Hkv) * half;
  const int e = blockIdx.x * 256 + threadIdx.x;
  auto val = [&](int n) { float a = bias ? bias[n] : 0.f; for (int k = 0; k < S; ++k) a += part[(int64_t)k * N + n]; return a; };
  if (e < n_pairs) {
    const int head = e / half, i = e - head * half;       // head < Hq: query head, else key head
    const int base = head * dh;                           // q and k sections are contiguous: [q heads | k heads]
    const float x0 = val(base + i), x1 = val(base + i + half);
    float sn, cs;
    sincosf((float)pos * inv_freq[i], &sn, &cs);
    const float y0 = x0 * cs - x1 * sn, y1 = x1 * cs + x0 * sn;
    if (head < Hq) { qout[base + i] = T::from_f32(y0); qout[base + i + half] = T::from_f32(y1); }
    else {
      uint16_t* k = kc + ((int64_t)(head - Hq) * max_ctx + pos) * dh;
      k[i] = T::from_f32(y0); k[i + half] = T::from_f32(y1);
    }
  } else if (e < n_pairs + Nk) {
    const int j = e - n_pairs, head = j / dh, d = j - head * dh;
    vc[((int64_t)head * max_ctx + pos) * dh + d] = T::from_f32(val(Nq + Nk + j));
  }
}

// grouped-query single-token attention: one workgroup per query head; DH/8 lanes share a key
template <typename T, int DH>
__global__ __launch_bounds__(256) void lm_attention(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
                                                    uint16_t* __restrict__ out, const LmState* __restrict__ st, int Hq, int Hkv, int max_ctx, float scale) {
  extern __shared__ float sc[];            // [max_ctx] scores, then red[4][DH] + red2[8]
  constexpr int LPK = DH / 8;              // lanes per key
  constexpr int KPW = 64 / LPK;            // keys per wave instruction
  float* red = sc + max_ctx;
  float* red2 = red + 4 * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, kvh = h / (Hq / Hkv);
  const int nk = st->pos + 1;
  const int c = lane % LPK, g = lane / LPK;
  const uint16_t* kb = kc + (int64_t)kvh * max_ctx * DH;
  const uint16_t* vb = vc + (int64_t)kvh * max_ctx * DH;
  float qf[8];
  {
    const s16x8 qv = *reinterpret_cast<const s16x8*>(q + h * DH + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = T::to_f32((uint16_t)qv[j]);
  }
  for (int k0 = wave * KPW * 4; k0 < nk; k0 += 4 * KPW * 4) {
    s16x8 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int key = k0 + KPW * u + g; key = key < nk ? key : nk - 1; kv[u] = *reinterpret_cast<const s16x8*>(kb + (int64_t)key * DH + c * 8); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += qf[j] * T::to_f32((uint16_t)kv[u][j]);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) dot += __shfl_xor(dot, o, 64);
      const int key = k0 + KPW * u + g;
      if (c == 0 && key < nk) sc[key] = dot * scale;
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int i = tid; i < nk; i += 256) m = fmaxf(m, sc[i]);
  m = wave_max(m);
  if (lane == 0) red2[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red2[0], red2[1]), fmaxf(red2[2], red2[3]));
  float sum = 0.f;
  for (int i = tid; i < nk; i += 256) { const float p = __expf(sc[i] - m); sc[i] = p; sum += p; }
  sum = wave_sum(sum);
  if (lane == 0) red2[4 + wave] = sum;
  __syncthreads();
  sum = (red2[4] + red2[5]) + (red2[6] + red2[7]);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k0 = wave * KPW * 4; k0 < nk; k0 += 4 * KPW * 4) {
    s16x8 vv[4]; float pw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + KPW * u + g; const int k2 = key < nk ? key : nk - 1;
      vv[u] = *reinterpret_cast<const s16x8*>(vb + (int64_t)k2 * DH + c * 8);
      pw[u] = key < nk ? sc[k2] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pw[u] * T::to_f32((uint16_t)vv[u][j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) acc[j] += __shfl_xor(acc[j], o, 64);
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave * DH + c * 8 + j] = acc[j];
  }
  __syncthreads();
  if (tid < DH) out[h * DH + tid] = T::from_f32(((red[tid] + red[DH + tid]) + (red[2 * DH + tid] + red[3 * DH + tid])) / sum);
}

// ---- sampler: repetition penalty -> temperature -> top-p (keep the first token that crosses p) -> inverse-CDF draw ----
// One workgroup of 1024 threads; the logits stay in HBM/L2 (V up to ~160 k).  top-p needs the descending order only to find
// the cut: a 3-level radix select over the fp32 bit pattern of the (max-shifted, unnormalised) probabilities finds the exact
// threshold value; ties at the threshold are kept lowest-index first.
__device__ __forceinline__ float blk1024_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ float blk1024_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 16; ++i) r = fmaxf(r, sh[i]);
  return r;
}

__global__ __launch_bounds__(1024) void lm_sample(float* __restrict__ logits, int V, int32_t* __restrict__ tokens, int32_t* __restrict__ hist,
                                                  const float* __restrict__ uniforms, LmState* __restrict__ st, mia_lm_sampler sp, int n_prompt, int max_ctx) {
  __shared__ float sh[16];
  __shared__ float hsum[2048];
  __shared__ unsigned hcnt[2048];
  __shared__ float s_f[4];
  __shared__ int s_i[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = st->pos;
  const int cur_len = pos + 1;
  if (cur_len < n_prompt || st->finished) {            // still consuming the prompt (or done): nothing to draw
    __syncthreads();
    if (tid == 0) st->pos = pos + 1;
    return;
  }
  const int n_hist = st->n_hist;
  // 1. repetition penalty over the last `rep_window` generated tokens (gather all, then scatter: duplicates penalised once)
  if (sp.rep_penalty != 1.0f && n_hist > 0) {
    float upd = 0.f; int tok = -1;
    if (tid < n_hist) { tok = hist[tid]; const float gth = logits[tok]; upd = gth < 0.f ? gth * sp.rep_penalty : gth / sp.rep_penalty; }
    __syncthreads();
    if (tid < n_hist) logits[tok] = upd;
    __syncthreads();
  }
  const float inv_t = 1.0f / fmaxf(sp.temperature, 1e-6f);
  // 2. softmax statistics of the temperature-scaled logits
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, logits[i] * inv_t);
  mx = blk1024_max(mx, sh);
  float tot = 0.f;
  for (int i = tid; i < V; i += 1024) tot += __expf(logits[i] * inv_t - mx);
  tot = blk1024_sum(tot, sh);
  // 3. top-p: find the bit pattern of the smallest kept (unnormalised) probability
  unsigned thr_bits = 0u;        // keep p > thr, plus `keep_ties` of the p == thr (lowest index first)
  int keep_ties = 0x7fffffff;
  float kept_sum = tot;
  const bool use_top_p = sp.top_p > 0.0f && sp.top_p < 1.0f && V > 1;
  if (use_top_p) {
    const float target = sp.top_p * tot;              // cumulative (descending) sum must EXCEED this
    unsigned prefix = 0u, mask = 0u;                  // bits fixed so far
    float cum_above = 0.f;                            // sum of probabilities strictly above the current candidate range
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int lvl = 0; lvl < 3; ++lvl) {
      for (int i = tid; i < 2048; i += 1024) { hsum[i] = 0.f; hcnt[i] = 0u; }
      __syncthreads();
      const int shf = shifts[lvl]; const unsigned nb = 1u << widths[lvl];
      for (int i = tid; i < V; i += 1024) {
        const float p = __expf(logits[i] * inv_t - mx);
        const unsigned b = __float_as_uint(p);
        if ((b & mask) != prefix) continue;
        const unsigned bin = (b >> shf) & (nb - 1);
        atomicAdd(&hsum[bin], p);
        atomicAdd(&hcnt[bin], 1u);
      }
      __syncthreads();
      if (tid == 0) {                                  // walk bins from the largest value down until the target is crossed
        float cum = cum_above; int sel = 0;
        for (int bin = (int)nb - 1; bin >= 0; --bin) {
          if (hcnt[bin] == 0u) continue;
          if (cum + hsum[bin] > target) { sel = bin; break; }
          cum += hsum[bin];
          sel = bin;                                   // (if never crossed: ends at the lowest occupied bin)
        }
        s_f[0] = cum; s_i[0] = sel;
      }
      __syncthreads();
      cum_above = s_f[0];
      prefix |= ((unsigned)s_i[0]) << shf;
      mask |= (nb - 1) << shf;
      __syncthreads();
    }
    thr_bits = prefix;
    const float thr = __uint_as_float(thr_bits);
    // ties: keep the smallest k >= 1 with cum_above + k*thr > target
    int k = 1;
    if (thr > 0.f) { const float need = (target - cum_above) / thr; k = (int)floorf(need) + 1; if (k < 1) k = 1; }
    keep_ties = k;
    // exact kept sum (threshold ties resolved by index order below)
    float ks = 0.f; unsigned nt = 0u;
    for (int i = tid; i < V; i += 1024) {
      const float p = __expf(logits[i] * inv_t - mx);
      const unsigned b = __float_as_uint(p);
      if (b > thr_bits) ks += p; else if (b == thr_bits) ++nt;
    }
    ks = blk1024_sum(ks, sh);
    const float ntf = blk1024_sum((float)nt, sh);
    if ((float)keep_ties > ntf) keep_ties = (int)ntf;
    kept_sum = ks + (float)keep_ties * thr;
  }
  // 4. inverse-CDF draw over the kept tokens in index order with the caller's uniform
  const float u = uniforms[st->n_gen];
  const float goal = u * kept_sum;
  // each wave owns a contiguous index range; lanes stride inside it (coalesced)
  const int per_wave = (V + 15) / 16;
  const int w_lo = wave * per_wave, w_hi = min(V, w_lo + per_wave);
  auto kept_p = [&](int i, int& tie_rank_before) -> float {   // tie handling needs the rank among ties: resolved in the ordered pass
    const float p = __expf(logits[i] * inv_t - mx);
    const unsigned b = __float_as_uint(p);
    if (!use_top_p || b > thr_bits) return p;
    if (b == thr_bits) { tie_rank_before = 1; return p; }
    return 0.f;
  };
  // ties at the threshold are rare (distinct floats); count ties per wave range to apply "lowest index first"
  float wsum = 0.f; int wties = 0;
  for (int i = w_lo + lane; i < w_hi; i += 64) { int t = 0; const float p = kept_p(i, t); if (t) ++wties; else wsum += p; }
  wsum = wave_sum(wsum);
  for (int o = 32; o > 0; o >>= 1) wties += __shfl_xor(wties, o, 64);
  __shared__ float wtot[16];
  __shared__ int wtie[16];
  if (lane == 0) { wtot[wave] = wsum; wtie[wave] = wties; }
  __syncthreads();
  if (tid == 0) {
    // serial over 16 ranges: find the range holding the goal (ties counted lowest-index first up to keep_ties)
    const float thr = __uint_as_float(thr_bits);
    float cum = 0.f; int ties_used = 0; int sel = 15;
    for (int w2 = 0; w2 < 16; ++w2) {
      const int tk = use_top_p ? min(wtie[w2], max(0, keep_ties - ties_used)) : 0;
      const float add = wtot[w2] + (float)tk * thr;
      if (cum + add > goal) { sel = w2; break; }
      cum += add; ties_used += tk;
    }
    s_f[1] = cum; s_i[1] = sel; s_i[2] = ties_used;
  }
  __syncthreads();
  if (wave == s_i[1]) {                                 // ordered scan of the selected range, 64 elements at a time
    const float thr = __uint_as_float(thr_bits);
    float cum = s_f[1]; int ties_used = s_i[2]; int found = -1;
    for (int base = w_lo; base < w_hi && found < 0; base += 64) {
      const int i = base + lane;
      float p = 0.f; int is_tie = 0;
      if (i < w_hi) { int t = 0; p = kept_p(i, t); is_tie = t; }
      // rank of tie lanes within this group
      const unsigned long long tmask = __ballot(is_tie != 0);
      const int rank = __popcll(tmask & ((1ull << lane) - 1ull));
      if (is_tie) p = (ties_used + rank < keep_ties) ? thr : 0.f;
      float incl = p;                                    // inclusive prefix over lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      const unsigned long long hit = __ballot(p > 0.f && cum + incl > goal);
      if (hit) found = base + (__ffsll((long long)hit) - 1);
      cum += __shfl(incl, 63, 64);
      ties_used += __popcll(tmask);
    }
    if (found < 0) {                                     // numerical corner: goal >= kept_sum; fall back to the last kept token of the range
      for (int i = w_hi - 1; i >= w_lo; --i) { int t = 0; if (kept_p(i, t) > 0.f) { found = i; break; } }
      if (found < 0) found = w_lo;
    }
    if (lane == 0) {
      const int next = found;
      const int ng = st->n_gen;
      if (cur_len < max_ctx) tokens[cur_len] = next;
      st->n_gen = ng + 1;
      bool stop = false;
      for (int k = 0; k < sp.n_stop; ++k) stop = stop || next == sp.stop_ids[k];
      if (!stop && sp.rep_window > 0) {                  // history is updated only for non-stop tokens (OrpheusTTS.swift:304-326)
        int nh = st->n_hist;
        if (nh < sp.rep_window) { hist[nh] = next; st->n_hist = nh + 1; }
        else { for (int k = 1; k < nh; ++k) hist[k - 1] = hist[k]; hist[nh - 1] = next; }
      }
      if (stop || ng + 1 >= sp.max_new_tokens || cur_len + 1 >= max_ctx) st->finished = 1;
      st->pos = pos + 1;
    }
  }
}

// ---- RAS sampler of CosyVoice2 (Qwen2LM.swift:295-321, 433-488): nucleus (top-p 0.8 capped at top-k 25, renormalised, drawn in
// descending-probability order); if the pick already occurs >= win*tau times among the last `win` emitted tokens, redraw from the
// full softmax; while i < min_len an EOS pick is rejected and the whole trial repeated (<= 100 times).  Every categorical draw is
// an inverse CDF with the next caller-provided uniform (u_cursor walks the stream).
struct RasParams { float top_p; int top_k; int win; float tau; int eos; int min_len; int max_len; int n_uniforms; };

__global__ __launch_bounds__(1024) void lm_sample_ras(const float* __restrict__ logits, int V, int32_t* __restrict__ tokens, int32_t* __restrict__ out_tokens,
                                                      const float* __restrict__ uniforms, LmState* __restrict__ st, RasParams rp, int max_ctx) {
  __shared__ float sh[16];
  __shared__ int shi[16];
  __shared__ float topv[32];
  __shared__ int topi[32];
  __shared__ int s_tok[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = st->pos, n_prompt = st->n_embeds;
  const int cur_len = pos + 1;
  if (cur_len < n_prompt || st->finished) { __syncthreads(); if (tid == 0 && !st->finished) st->pos = pos + 1; return; }
  const int step_i = cur_len - n_prompt;                 // loop index i of inferenceLoop
  // softmax statistics
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, logits[i]);
  mx = blk1024_max(mx, sh);
  float tot = 0.f;
  for (int i = tid; i < V; i += 1024) tot += __expf(logits[i] - mx);
  tot = blk1024_sum(tot, sh);
  // top-k by iterated block argmax (k <= 32): (value, index) with lowest index on ties; previously taken entries are skipped
  const int K = rp.top_k < 32 ? rp.top_k : 32;
  for (int r = 0; r < K; ++r) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 1024) {
      const float v = logits[i];
      bool taken = false;
      for (int q = 0; q < r; ++q) taken = taken || topi[q] == i;
      if (!taken && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sh[wave] = bv; shi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = sh[0]; int ix = shi[0];
      for (int w2 = 1; w2 < 16; ++w2) if (sh[w2] > v || (sh[w2] == v && shi[w2] < ix)) { v = sh[w2]; ix = shi[w2]; }
      topv[r] = __expf(v - mx) / tot; topi[r] = ix;
    }
    __syncthreads();
  }
  // per-wave partial sums of the full softmax in index order (for the fallback draw)
  const int per_wave = (V + 15) / 16;
  const int w_lo = wave * per_wave, w_hi = min(V, w_lo + per_wave);
  float wsum = 0.f;
  for (int i = w_lo + lane; i < w_hi; i += 64) wsum += __expf(logits[i] - mx);
  wsum = wave_sum(wsum);
  __shared__ float wtot[16];
  if (lane == 0) wtot[wave] = wsum;
  __syncthreads();
  __shared__ int s_need_full; __shared__ float s_goal; __shared__ int s_sel; __shared__ float s_cum;
  int trials = 0;
  int pick = -1;
  while (true) {                                         // trial loop (uniform across the block through shared state)
    if (tid == 0) {
      int cur = st->u_cursor;
      // nucleus: n = min(count(cumsum < top_p) + 1, top_k)
      float cum = 0.f; int below = 0;
      for (int r = 0; r < K; ++r) { cum += topv[r]; if (cum < rp.top_p) ++below; }
      int n = below + 1; if (n > K) n = K;
      float ns = 0.f;
      for (int r = 0; r < n; ++r) ns += topv[r];
      const float u = uniforms[cur < rp.n_uniforms ? cur : rp.n_uniforms - 1]; ++cur;
      float c2 = 0.f; int sel = n - 1;
      for (int r = 0; r < n; ++r) { c2 += topv[r]; if (c2 > u * ns) { sel = r; break; } }
      int tok = topi[sel];
      // repetition-aware fallback (rasSampling :463-488): over the last `win` EMITTED tokens
      int rep = 0; const int no = st->n_out;
      for (int q = max(0, no - rp.win); q < no; ++q) rep += out_tokens[q] == tok;
      int need_full = 0;
      if (no > 0 && (float)rep >= (float)rp.win * rp.tau) {
        need_full = 1;
        const float u2 = uniforms[cur < rp.n_uniforms ? cur : rp.n_uniforms - 1]; ++cur;
        const float goal = u2 * tot;
        float c3 = 0.f; int ws = 15;
        for (int w2 = 0; w2 < 16; ++w2) { if (c3 + wtot[w2] > goal) { ws = w2; break; } c3 += wtot[w2]; }
        s_goal = goal; s_sel = ws; s_cum = c3;
      }
      st->u_cursor = cur;
      s_need_full = need_full; s_tok[0] = tok;
    }
    __syncthreads();
    if (s_need_full) {
      if (wave == s_sel) {                               // ordered scan of the selected index range
        float cum = s_cum; int found = -1;
        for (int base = w_lo; base < w_hi && found < 0; base += 64) {
          const int i = base + lane;
          const float p = i < w_hi ? __expf(logits[i] - mx) : 0.f;
          float incl = p;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
          const unsigned long long hit = __ballot(p > 0.f && cum + incl > s_goal);
          if (hit) found = base + (__ffsll((long long)hit) - 1);
          cum += __shfl(incl, 63, 64);
        }
        if (found < 0) found = w_hi - 1;
        if (lane == 0) s_tok[0] = found;
      }
      __syncthreads();
    }
    pick = s_tok[0];
    ++trials;
    const bool ignore_eos = step_i < rp.min_len;
    if (!(ignore_eos && pick == rp.eos) || trials > 100) break;   // the Swift throws after 100 rejected trials; we keep the EOS
    __syncthreads();
  }
  if (tid == 0) {
    const int ng = st->n_gen;
    st->n_gen = ng + 1;
    if (pick == rp.eos) st->finished = 1;
    else {
      if (cur_len < max_ctx) tokens[cur_len] = pick;     // embedding input of the next step (speech_embedding[pick])
      if (pick < rp.eos) { out_tokens[st->n_out] = pick; st->n_out += 1; }   // ids above EOS (fill tokens) are fed back, not emitted
      if (step_i + 1 >= rp.max_len || cur_len + 1 >= max_ctx) st->finished = 1;
      st->pos = pos + 1;
    }
  }
}

// greedy / plain path: advance only (logits are read back by the host)
__global__ void lm_advance(LmState* st) { st->pos += 1; }

}  // namespace

// ---- host side ------------------------------------------------------------------------------------
namespace {

struct LmLoader {
  mia_lm* m;
  std::map<std::string, const mia_tensor_view*> by_name;
  std::string err;
  const mia_tensor_view* find(const std::string& n, bool req = true) {
    auto it = by_name.find(n);
    if (it == by_name.end()) { if (req && err.empty()) err = "missing tensor '" + n + "'"; return nullptr; }
    return it->second;
  }
  static float h2f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
  bool to_f32(const std::string& n, std::vector<float>& out, int64_t rows, int64_t cols, bool req = true) {
    const mia_tensor_view* t = find(n, req);
    if (!t) return false;
    const bool ok = cols > 0 ? (t->ndim == 2 && t->shape[0] == rows && t->shape[1] == cols) : (t->ndim == 1 && t->shape[0] == rows);
    if (!ok) { if (err.empty()) err = "tensor '" + n + "' has an unexpected shape"; return false; }
    const int64_t numel = rows * (cols > 0 ? cols : 1);
    out.resize(numel);
    if (t->dtype == MIA_F32) memcpy(out.data(), t->data, numel * 4);
    else if (t->dtype == MIA_F16) { const uint16_t* p = (const uint16_t*)t->data; for (int64_t i = 0; i < numel; ++i) out[i] = h2f(p[i]); }
    else { const uint16_t* p = (const uint16_t*)t->data; for (int64_t i = 0; i < numel; ++i) { uint32_t u = (uint32_t)p[i] << 16; memcpy(&out[i], &u, 4); } }
    return true;
  }
  void* dev(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes + 64) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    m->allocs.push_back(p);
    return p;
  }
  float* up32(const std::vector<float>& v) { float* d = (float*)dev(v.size() * 4); if (d) (void)hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; }
  void* up16(const std::vector<float>& v) {
    std::vector<uint16_t> q(v.size());
    if (m->dtype == MIA_F16) for (size_t i = 0; i < v.size(); ++i) { _Float16 hh = (_Float16)v[i]; memcpy(&q[i], &hh, 2); }
    else for (size_t i = 0; i < v.size(); ++i) { uint32_t u; memcpy(&u, &v[i], 4); u += 0x7fffu + ((u >> 16) & 1); q[i] = (uint16_t)(u >> 16); }
    void* d = dev(q.size() * 2);
    if (d) (void)hipMemcpy(d, q.data(), q.size() * 2, hipMemcpyHostToDevice);
    return d;
  }
};

int pick_split(int K, int want) { for (int s = want; s > 1; --s) if (K % (32 * s) == 0) return s; return 1; }

int lm_enqueue_step(mia_lm* m, bool sampling, const mia_lm_sampler& sp, int n_prompt, const RasParams* ras = nullptr) {
  hipStream_t s = m->ctx->stream;
  const mia_lm_config& c = m->cfg;
  const int D = c.hidden, dh = c.head_dim, Nq = c.n_heads * dh, Nk = c.n_kv_heads * dh, Nqkv = Nq + 2 * Nk;
  const bool f16 = m->dtype == MIA_F16;
  auto skinny = [&](const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldo, int N, int K, int S, int mode) {
    SkinnyArgs a{(const uint16_t*)A, lda, (const uint16_t*)W, bias, out, ldo, nullptr, nullptr, nullptr, 1, N, K, S, MIA_ACT_NONE, 0, 0, 0};
    return skinny_gemm_launch(a, mode, m->dtype, s);
  };
#define LAUNCH_T(kern, grid, block, lds, ...) do { if (f16) hipLaunchKernelGGL((kern<F16>), grid, block, lds, s, __VA_ARGS__); else hipLaunchKernelGGL((kern<BF16>), grid, block, lds, s, __VA_ARGS__); } while (0)
  LAUNCH_T(lm_embed_norm, dim3(1), dim3(256), 0, m->tokens, (const uint16_t*)m->embed, (const uint16_t*)m->gen_embed, m->embeds, m->layers[0].in_norm, m->x, (uint16_t*)m->h, m->state, D, c.rms_eps);
  const size_t att_lds = (size_t)(c.max_ctx + 4 * dh + 8) * 4;
  for (int l = 0; l < c.n_layers; ++l) {
    const LmLayer& L = m->layers[l];
    uint16_t* kc = (uint16_t*)m->k_cache + (size_t)l * c.n_kv_heads * c.max_ctx * dh;
    uint16_t* vc = (uint16_t*)m->v_cache + (size_t)l * c.n_kv_heads * c.max_ctx * dh;
    if (skinny(m->h, D, L.wqkv, nullptr, m->qkv_part, 0, Nqkv, D, m->S_qkv, SK_PARTIAL)) return -1;
    const int n_el = (c.n_heads + c.n_kv_heads) * (dh / 2) + Nk;
    LAUNCH_T(lm_rope_cache, dim3((n_el + 255) / 256), dim3(256), 0, m->qkv_part, m->S_qkv, L.bqkv, m->inv_freq, (uint16_t*)m->q, kc, vc, m->state,
             c.n_heads, c.n_kv_heads, dh, c.max_ctx);
    const float scale = 1.0f / sqrtf((float)dh);
    if (dh == 128) {
      if (f16) hipLaunchKernelGGL((lm_attention<F16, 128>), dim3(c.n_heads), dim3(256), att_lds, s, (const uint16_t*)m->q, kc, vc, (uint16_t*)m->att, m->state, c.n_heads, c.n_kv_heads, c.max_ctx, scale);
      else hipLaunchKernelGGL((lm_attention<BF16, 128>), dim3(c.n_heads), dim3(256), att_lds, s, (const uint16_t*)m->q, kc, vc, (uint16_t*)m->att, m->state, c.n_heads, c.n_kv_heads, c.max_ctx, scale);
    } else {
      if (f16) hipLaunchKernelGGL((lm_attention<F16, 64>), dim3(c.n_heads), dim3(256), att_lds, s, (const uint16_t*)m->q, kc, vc, (uint16_t*)m->att, m->state, c.n_heads, c.n_kv_heads, c.max_ctx, scale);
      else hipLaunchKernelGGL((lm_attention<BF16, 64>), dim3(c.n_heads), dim3(256), att_lds, s, (const uint16_t*)m->q, kc, vc, (uint16_t*)m->att, m->state, c.n_heads, c.n_kv_heads, c.max_ctx, scale);
    }
    if (skinny(m->att, Nq, L.wo, nullptr, m->partial, 0, D, Nq, m->S_o, SK_PARTIAL)) return -1;
    LAUNCH_T(lm_reduce_norm, dim3(1), dim3(256), 0, m->partial, m->S_o, L.post_norm, m->x, (uint16_t*)m->h, D, c.rms_eps);
    if (skinny(m->h, D, L.wgu, nullptr, m->act, c.inter, 2 * c.inter, D, 1, SK_SWIGLU)) return -1;
    if (skinny(m->act, c.inter, L.wdown, nullptr, m->partial, 0, D, c.inter, m->S_down, SK_PARTIAL)) return -1;
    LAUNCH_T(lm_reduce_norm, dim3(1), dim3(256), 0, m->partial, m->S_down, l + 1 < c.n_layers ? m->layers[l + 1].in_norm : m->final_norm, m->x, (uint16_t*)m->h, D, c.rms_eps);
  }
#undef LAUNCH_T
  const int HV = m->head_vocab > 0 ? m->head_vocab : c.vocab;
  if (skinny(m->h, D, m->lm_head, m->head_bias, m->logits, HV, HV, D, 1, SK_OUTF32)) return -1;
  if (ras) hipLaunchKernelGGL(lm_sample_ras, dim3(1), dim3(1024), 0, s, m->logits, HV, m->tokens, m->out_tokens, m->uniforms, m->state, *ras, c.max_ctx);
  else if (sampling) hipLaunchKernelGGL(lm_sample, dim3(1), dim3(1024), 0, s, m->logits, HV, m->tokens, m->hist, m->uniforms, m->state, sp, n_prompt, c.max_ctx);
  else hipLaunchKernelGGL(lm_advance, dim3(1), dim3(1), 0, s, m->state);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int lm_graph(mia_lm* m, bool sampling, const mia_lm_sampler& sp, int n_prompt) {
  mia_ctx* ctx = m->ctx;
  static const bool no_graph = getenv("MIA_NO_GRAPH") != nullptr;
  if (no_graph) return 1;
  mia_lm_sampler key = sp; key.max_new_tokens = sampling ? sp.max_new_tokens : 0;
  // n_prompt is a kernel argument of the sampler: fold it into the key via top_k-unused field
  if (m->graph_valid && m->graph_sampling == sampling && memcmp(&m->graph_sampler, &key, sizeof(key)) == 0 && m->graph_sampler.reserved == n_prompt) return 0;
  if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
  hipGraph_t g = nullptr;
  MIA_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  const int erc = lm_enqueue_step(m, sampling, sp, n_prompt);
  hipError_t ce = hipStreamEndCapture(ctx->stream, &g);
  if (erc != 0 || ce != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return mia_fail(ctx, MIA_ERR_DEVICE, "lm: step graph capture failed"); }
  hipError_t ie = hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess) return mia_fail(ctx, MIA_ERR_DEVICE, "lm: hipGraphInstantiate failed");
  m->graph_sampler = key; m->graph_sampler.reserved = n_prompt; m->graph_sampling = sampling; m->graph_valid = true;
  return 0;
}

}  // namespace

extern "C" void mia_lm_free(mia_lm* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->graph) (void)hipGraphExecDestroy(m->graph);
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

extern "C" mia_lm* mia_lm_load(mia_ctx* ctx, const mia_lm_config* cfg, const mia_tensor_view* tensors, int n_tensors, int dtype) {
  if (!ctx) return nullptr;
  auto fail = [&](mia_lm* m, const std::string& msg) -> mia_lm* { ctx->err = "lm_load: " + msg; if (m) mia_lm_free(m); return nullptr; };
  if (!cfg || !tensors || n_tensors <= 0) return fail(nullptr, "null arguments");
  if (dtype != MIA_BF16 && dtype != MIA_F16) return fail(nullptr, "dtype must be MIA_BF16 or MIA_F16");
  const mia_lm_config& c = *cfg;
  if (c.head_dim != 64 && c.head_dim != 128) return fail(nullptr, "head_dim must be 64 or 128");
  if (c.hidden % 32 || c.hidden > 4096 || c.inter % 32 || c.n_heads % c.n_kv_heads || c.vocab <= 0 || c.n_layers <= 0 || c.max_ctx <= 0 || c.max_ctx > 8192)
    return fail(nullptr, "unsupported dimensions (hidden <= 4096 and % 32, inter % 32, max_ctx <= 8192)");
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(nullptr, "hipSetDevice failed");
  mia_lm* m = new mia_lm(); m->ctx = ctx; m->cfg = c; m->dtype = dtype;
  LmLoader L; L.m = m;
  for (int i = 0; i < n_tensors; ++i) if (tensors[i].name && tensors[i].data) L.by_name[tensors[i].name] = &tensors[i];
  const int D = c.hidden, dh = c.head_dim, Nq = c.n_heads * dh, Nk = c.n_kv_heads * dh;
  std::vector<float> t, t2, t3;
  if (L.to_f32("model.embed_tokens.weight", t, c.vocab, D)) m->embed = L.up16(t);
  if (c.tie_embeddings) m->lm_head = m->embed;
  else if (L.find("lm_head.weight", false) && L.to_f32("lm_head.weight", t, c.vocab, D)) m->lm_head = L.up16(t);
  if (L.to_f32("model.norm.weight", t, D, 0)) m->final_norm = L.up32(t);
  if (const mia_tensor_view* hv = L.find("llm_decoder.weight", false)) {      // Qwen2LM: separate output head + speech embedding
    if (hv->ndim == 2 && hv->shape[1] == D) {
      m->head_vocab = (int)hv->shape[0];
      if (L.to_f32("llm_decoder.weight", t, m->head_vocab, D)) m->lm_head = L.up16(t);
      if (L.find("llm_decoder.bias", false) && L.to_f32("llm_decoder.bias", t, m->head_vocab, 0)) m->head_bias = L.up32(t);
    } else L.err = "llm_decoder.weight has an unexpected shape";
  }
  if (const mia_tensor_view* gv = L.find("speech_embedding.weight", false)) {
    if (gv->ndim == 2 && gv->shape[1] == D) { m->gen_rows = (int)gv->shape[0]; if (L.to_f32("speech_embedding.weight", t, m->gen_rows, D)) m->gen_embed = L.up16(t); }
    else L.err = "speech_embedding.weight has an unexpected shape";
  }
  {  // rotary inverse frequencies: plain RoPE(base) or Llama3RoPE (Llama3RoPE.swift:41-65: period-like `freqs`, MLX divides positions by them)
    std::vector<float> inv(dh / 2);
    for (int i = 0; i < dh / 2; ++i) {
      float freq = powf(c.rope_theta, (float)(2 * i) / (float)dh);
      if (c.rope_llama3) {
        const float low_wl = (float)c.rope_old_ctx / c.rope_low, high_wl = (float)c.rope_old_ctx / c.rope_high;
        const float wl = 2.0f * (float)M_PI * freq;
        float f = wl > low_wl ? freq * c.rope_factor : freq;
        if (wl > high_wl && wl < low_wl) {
          const float smooth = ((float)c.rope_old_ctx / wl - c.rope_low) / (c.rope_high - c.rope_low);
          f = f / ((1.0f - smooth) / c.rope_factor + smooth);
        }
        freq = f;
      }
      inv[i] = 1.0f / freq;
    }
    m->inv_freq = L.up32(inv);
  }
  m->layers.resize(c.n_layers);
  for (int l = 0; l < c.n_layers && L.err.empty(); ++l) {
    const std::string p = "model.layers." + std::to_string(l);
    LmLayer& ly = m->layers[l];
    if (L.to_f32(p + ".input_layernorm.weight", t, D, 0)) ly.in_norm = L.up32(t);
    if (L.to_f32(p + ".post_attention_layernorm.weight", t, D, 0)) ly.post_norm = L.up32(t);
    std::vector<float> qkv((size_t)(Nq + 2 * Nk) * D);
    if (L.to_f32(p + ".self_attn.q_proj.weight", t, Nq, D) && L.to_f32(p + ".self_attn.k_proj.weight", t2, Nk, D) && L.to_f32(p + ".self_attn.v_proj.weight", t3, Nk, D)) {
      memcpy(qkv.data(), t.data(), t.size() * 4); memcpy(qkv.data() + t.size(), t2.data(), t2.size() * 4); memcpy(qkv.data() + t.size() + t2.size(), t3.data(), t3.size() * 4);
      ly.wqkv = L.up16(qkv);
    }
    if (c.qkv_bias) {
      std::vector<float> b((size_t)Nq + 2 * Nk);
      if (L.to_f32(p + ".self_attn.q_proj.bias", t, Nq, 0) && L.to_f32(p + ".self_attn.k_proj.bias", t2, Nk, 0) && L.to_f32(p + ".self_attn.v_proj.bias", t3, Nk, 0)) {
        memcpy(b.data(), t.data(), t.size() * 4); memcpy(b.data() + Nq, t2.data(), t2.size() * 4); memcpy(b.data() + Nq + Nk, t3.data(), t3.size() * 4);
        ly.bqkv = L.up32(b);
      }
    }
    if (L.to_f32(p + ".self_attn.o_proj.weight", t, D, Nq)) ly.wo = L.up16(t);
    if (L.to_f32(p + ".mlp.gate_proj.weight", t, c.inter, D) && L.to_f32(p + ".mlp.up_proj.weight", t2, c.inter, D)) {
      std::vector<float> gu((size_t)2 * c.inter * D);
      for (int r = 0; r < c.inter; ++r) { memcpy(&gu[(size_t)(2 * r) * D], &t[(size_t)r * D], (size_t)D * 4); memcpy(&gu[(size_t)(2 * r + 1) * D], &t2[(size_t)r * D], (size_t)D * 4); }
      ly.wgu = L.up16(gu);
    }
    if (L.to_f32(p + ".mlp.down_proj.weight", t, D, c.inter)) ly.wdown = L.up16(t);
  }
  if (!L.err.empty()) return fail(m, L.err);
  m->S_qkv = pick_split(D, 4); m->S_o = pick_split(Nq, 4); m->S_down = pick_split(c.inter, 8);
  const size_t kv = (size_t)c.n_layers * c.n_kv_heads * c.max_ctx * dh * 2;
  m->k_cache = L.dev(kv); m->v_cache = L.dev(kv);
  m->x = (float*)L.dev((size_t)D * 4); m->h = L.dev((size_t)D * 2);
  m->qkv_part = (float*)L.dev((size_t)4 * (Nq + 2 * Nk) * 4); m->q = L.dev((size_t)Nq * 2); m->att = L.dev((size_t)Nq * 2); m->act = L.dev((size_t)c.inter * 2);
  m->partial = (float*)L.dev((size_t)8 * D * 4); m->logits = (float*)L.dev((size_t)std::max(c.vocab, m->head_vocab) * 4);
  m->tokens = (int32_t*)L.dev((size_t)c.max_ctx * 4); m->hist = (int32_t*)L.dev(64 * 4); m->uniforms = (float*)L.dev((size_t)c.max_ctx * 4);
  m->state = (LmState*)L.dev(sizeof(LmState));
  m->embeds = (float*)L.dev((size_t)c.max_ctx * D * 4); m->out_tokens = (int32_t*)L.dev((size_t)c.max_ctx * 4);
  if (!L.err.empty()) return fail(m, L.err);
  (void)hipMemset(m->k_cache, 0, kv); (void)hipMemset(m->v_cache, 0, kv); (void)hipMemset(m->state, 0, sizeof(LmState));
  if (hipDeviceSynchronize() != hipSuccess) return fail(m, "device error during upload");
  return m;
}

extern "C" int mia_lm_reset(mia_lm* m) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_HIP(m->ctx, hipSetDevice(m->ctx->device));
  MIA_HIP(m->ctx, hipMemsetAsync(m->state, 0, sizeof(LmState), m->ctx->stream));
  return MIA_OK;
}

// Feed n tokens (appended at the current position) and return the logits after the last one (model(ids, cache) + [0,-1]).
extern "C" int mia_lm_forward(mia_lm* m, const int32_t* ids, int n, float* last_logits) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, ids && n > 0, "lm_forward: ids required");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  LmState st{};
  MIA_HIP(ctx, hipMemcpyAsync(&st, m->state, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MIA_CHECK_ARG(ctx, st.pos + n <= m->cfg.max_ctx, "lm_forward: context overflow (%d + %d > %d)", st.pos, n, m->cfg.max_ctx);
  for (int i = 0; i < n; ++i) MIA_CHECK_ARG(ctx, ids[i] >= 0 && ids[i] < m->cfg.vocab, "lm_forward: token %d out of vocabulary", ids[i]);
  MIA_HIP(ctx, hipMemcpyAsync(m->tokens + st.pos, ids, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  mia_lm_sampler none{};
  const int gr = lm_graph(m, false, none, 0);
  if (gr < 0) return gr;
  for (int i = 0; i < n; ++i) {
    if (gr == 0) MIA_HIP(ctx, hipGraphLaunch(m->graph, ctx->stream));
    else if (lm_enqueue_step(m, false, none, 0)) return mia_fail(ctx, MIA_ERR_DEVICE, "lm_forward: launch failed");
  }
  if (last_logits) MIA_HIP(ctx, hipMemcpyAsync(last_logits, m->logits, (size_t)m->cfg.vocab * 4, hipMemcpyDeviceToHost, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIA_OK;
}

// prompt -> sampled continuation, all on device (OrpheusTTS.generateChunk's loop, OrpheusTTS.swift:245-348).
extern "C" int mia_lm_generate(mia_lm* m, const int32_t* prompt, int n_prompt, const mia_lm_sampler* sp, const float* uniforms,
                               int32_t* out_tokens, int32_t* n_out) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, prompt && n_prompt > 0 && sp && uniforms && out_tokens && n_out, "lm_generate: null arguments");
  MIA_CHECK_ARG(ctx, sp->max_new_tokens > 0 && n_prompt + sp->max_new_tokens <= m->cfg.max_ctx, "lm_generate: prompt + max_new_tokens exceeds max_ctx");
  MIA_CHECK_ARG(ctx, sp->rep_window >= 0 && sp->rep_window <= 64 && sp->n_stop >= 0 && sp->n_stop <= 4, "lm_generate: rep_window <= 64, n_stop <= 4");
  for (int i = 0; i < n_prompt; ++i) MIA_CHECK_ARG(ctx, prompt[i] >= 0 && prompt[i] < m->cfg.vocab, "lm_generate: token %d out of vocabulary", prompt[i]);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  MIA_HIP(ctx, hipMemsetAsync(m->state, 0, sizeof(LmState), s));
  MIA_HIP(ctx, hipMemcpyAsync(m->tokens, prompt, (size_t)n_prompt * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(m->uniforms, uniforms, (size_t)sp->max_new_tokens * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  const int gr = lm_graph(m, true, *sp, n_prompt);
  if (gr < 0) return gr;
  const int total = n_prompt + sp->max_new_tokens - 1;
  LmState st{};
  for (int step = 0; step < total; ++step) {
    if (gr == 0) MIA_HIP(ctx, hipGraphLaunch(m->graph, s));
    else if (lm_enqueue_step(m, true, *sp, n_prompt)) return mia_fail(ctx, MIA_ERR_DEVICE, "lm_generate: launch failed");
    if (step >= n_prompt && (step & 15) == 15) {
      MIA_HIP(ctx, hipMemcpyAsync(&st, m->state, sizeof(st), hipMemcpyDeviceToHost, s));
      MIA_HIP(ctx, hipStreamSynchronize(s));
      if (st.finished) break;
    }
  }
  MIA_HIP(ctx, hipMemcpyAsync(&st, m->state, sizeof(st), hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  *n_out = st.n_gen;
  MIA_HIP(ctx, hipMemcpyAsync(out_tokens, m->tokens + n_prompt, (size_t)st.n_gen * 4, hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}

// standalone sampler on caller-provided logits (OrpheusTTS.sampleNextToken, OrpheusTTS.swift:375-470)
extern "C" int mia_sample_top_p(mia_ctx* ctx, const float* logits, int V, const int32_t* history, int n_hist, float rep_penalty, float temperature,
                                float top_p, float uniform, int32_t* out) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, logits && out && V > 0 && n_hist >= 0 && n_hist <= 64, "sample_top_p: bad arguments");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const size_t need = align_up((size_t)V * 4, 256) + 1024;
  char* ws = (char*)mia_workspace(ctx, need);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_logits = (float*)ws;
  char* tail = ws + align_up((size_t)V * 4, 256);
  int32_t* d_tok = (int32_t*)tail;            // [2]: tokens[0..1]
  int32_t* d_hist = (int32_t*)(tail + 64);    // [64]
  float* d_u = (float*)(tail + 64 + 256);
  LmState* d_st = (LmState*)(tail + 64 + 256 + 64);
  LmState st{0, n_hist, 0, 0};
  hipStream_t s = ctx->stream;
  MIA_HIP(ctx, hipMemcpyAsync(d_logits, logits, (size_t)V * 4, hipMemcpyHostToDevice, s));
  if (n_hist) MIA_HIP(ctx, hipMemcpyAsync(d_hist, history, (size_t)n_hist * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_u, &uniform, 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_st, &st, sizeof(st), hipMemcpyHostToDevice, s));
  mia_lm_sampler sp{}; sp.temperature = temperature; sp.top_p = top_p; sp.rep_penalty = rep_penalty; sp.rep_window = 0; sp.max_new_tokens = 1;
  hipLaunchKernelGGL(lm_sample, dim3(1), dim3(1024), 0, s, d_logits, V, d_tok, d_hist, d_u, d_st, sp, 1, 2);
  MIA_HIP(ctx, hipGetLastError());
  MIA_HIP(ctx, hipMemcpyAsync(out, d_tok + 1, 4, hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}

// Qwen2LM.inference / inferenceLoop (TTS/CosyVoice2/LLM/Qwen2LM.swift:335-427) on device: prompt given as embedding rows
// [sos, text..., task, prompt speech...] (the caller gathers them from its three tables), generated ids embedded through
// speech_embedding, logits through llm_decoder, RAS sampling with explicit uniforms.
extern "C" int mia_lm_generate_ras(mia_lm* m, const float* prompt_embeds, int n_prompt, const mia_ras_params* rp, const float* uniforms, int n_uniforms,
                                   int32_t* out_tokens, int32_t* n_out) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, prompt_embeds && n_prompt > 0 && rp && uniforms && n_uniforms > 0 && out_tokens && n_out, "lm_generate_ras: null arguments");
  MIA_CHECK_ARG(ctx, m->gen_embed && m->head_vocab > 0, "lm_generate_ras: model has no speech_embedding / llm_decoder tensors");
  MIA_CHECK_ARG(ctx, rp->max_len > 0 && n_prompt + rp->max_len <= m->cfg.max_ctx, "lm_generate_ras: prompt + max_len exceeds max_ctx");
  MIA_CHECK_ARG(ctx, rp->top_k > 0 && rp->top_k <= 32 && rp->win >= 0 && rp->win <= 64 && rp->eos >= 0 && rp->eos < m->head_vocab, "lm_generate_ras: bad sampler parameters");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  LmState st{}; st.n_embeds = n_prompt;
  MIA_HIP(ctx, hipMemcpyAsync(m->state, &st, sizeof(st), hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(m->embeds, prompt_embeds, (size_t)n_prompt * m->cfg.hidden * 4, hipMemcpyHostToDevice, s));
  const int nu = std::min(n_uniforms, m->cfg.max_ctx);
  MIA_HIP(ctx, hipMemcpyAsync(m->uniforms, uniforms, (size_t)nu * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  RasParams r{rp->top_p, rp->top_k, rp->win, rp->tau, rp->eos, rp->min_len, rp->max_len, nu};
  mia_lm_sampler none{};
  const int total = n_prompt + rp->max_len - 1;
  for (int step = 0; step < total; ++step) {
    if (lm_enqueue_step(m, true, none, n_prompt, &r)) return mia_fail(ctx, MIA_ERR_DEVICE, "lm_generate_ras: launch failed");
    if (step >= n_prompt && (step & 15) == 15) {
      MIA_HIP(ctx, hipMemcpyAsync(&st, m->state, sizeof(st), hipMemcpyDeviceToHost, s));
      MIA_HIP(ctx, hipStreamSynchronize(s));
      if (st.finished) break;
    }
  }
  MIA_HIP(ctx, hipMemcpyAsync(&st, m->state, sizeof(st), hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  *n_out = st.n_out;
  MIA_HIP(ctx, hipMemcpyAsync(out_tokens, m->out_tokens, (size_t)st.n_out * 4, hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}
