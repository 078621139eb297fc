// decode_kernels.hip -- the per-token kernels of the batched Whisper greedy decoder (gfx950).
//
// Replaces, per step, TextDecoder.callAsFunction (STT/Whisper/Layers/TextDecoder.swift:53-96), the decoder half of
// ResidualAttentionBlock (ResidualAttentionBlock.swift:51-95), WhisperMultiHeadAttention with KV cache
// (MultiHeadAttention.swift:40-135) and the logit rules + argmax of GreedyDecoder.decode
// (STT/Whisper/WhisperDecoding.swift:186-358).  All state (position, tokens, rule state, log-prob sums) lives in
// HBM, every kernel reads the position from DecState, so one captured hipGraph replays for every step and the
// host never synchronises per token (the reference does >= 4 .item() syncs per token).
//
// Roofline: a decode step is HBM-bound -- weights are streamed once (skinny MFMA GEMM, M <= 32 rows), the
// cross-attention K/V of every clip is streamed once (dec_attention).
#include "decode.h"

// ------------------------------------------------------------------------------------------------
// x[b] = E[token[b][pos]] + P[pos];  h[b] = LN(x[b])          (TextDecoder.swift:67 + first attn_ln)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void dec_embed_ln(const int32_t* __restrict__ tokens, const uint16_t* __restrict__ emb,
                                                   const float* __restrict__ pos_emb, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ x,
                                                   uint16_t* __restrict__ h, const DecState* __restrict__ st, int D, int n_ctx) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int pos = st->pos;
  const int tok = tokens[b * n_ctx + pos];
  const uint16_t* e = emb + (int64_t)tok * D;
  const float* p = pos_emb + (int64_t)pos * D;
  float* xr = x + (int64_t)b * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float v = T::to_f32(e[i]) + p[i];
    xr[i] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
  for (int i = lane; i < D; i += 64) h[(int64_t)b * D + i] = T::from_f32((xr[i] - mean) * rstd * gamma[i] + beta[i]);
}

// ------------------------------------------------------------------------------------------------
// x[b] += bias + sum_s partial[s][b];  h[b] = LN(x[b])        (residual add + next LayerNorm, fixed-order split-K sum)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void dec_reduce_ln(const float* __restrict__ partial, int S, int B, const float* __restrict__ bias,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ x, uint16_t* __restrict__ h, int D) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* xr = x + (int64_t)b * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    float v = xr[i] + bias[i];
    for (int k = 0; k < S; ++k) v += partial[((int64_t)k * B + b) * D + i];
    xr[i] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xr[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
  for (int i = lane; i < D; i += 64) h[(int64_t)b * D + i] = T::from_f32((xr[i] - mean) * rstd * gamma[i] + beta[i]);
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM for decode (M <= 32 rows per block-z): out[m][n] = sum_k A[m][k] W[n][k].
// One wave per 16 output columns and K-slice; weights go HBM -> VGPR (each weight byte is read exactly once),
// activations come from L2; v_mfma_f32_16x16x32 with W as the row operand so a lane owns 4 consecutive n.
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(64) void dec_skinny_gemm(SkinnyArgs a) {
  const int lane = threadIdx.x;
  const int n0 = blockIdx.x * 16;
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * 32;
  const int Kc = a.K / a.S;
  const int kbeg = split * Kc;
  const int r = lane & 15, c = lane >> 4;
  int wn = n0 + r; wn = wn < a.N ? wn : a.N - 1;
  int am0 = m0 + r; am0 = am0 < a.M ? am0 : a.M - 1;
  int am1 = m0 + 16 + r; am1 = am1 < a.M ? am1 : a.M - 1;
  const uint16_t* wp = a.W + (int64_t)wn * a.K + kbeg + 8 * c;
  const uint16_t* ap0 = a.A + (int64_t)am0 * a.lda + kbeg + 8 * c;
  const uint16_t* ap1 = a.A + (int64_t)am1 * a.lda + kbeg + 8 * c;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // 4 K-steps (128 k) per iteration: 12 independent 16-byte loads in flight before the first MFMA
  int k = 0;
  for (; k + 128 <= Kc; k += 128) {
    s16x8 fw[4], fa0[4], fa1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fw[u] = *reinterpret_cast<const s16x8*>(wp + k + 32 * u);
      fa0[u] = *reinterpret_cast<const s16x8*>(ap0 + k + 32 * u);
      fa1[u] = *reinterpret_cast<const s16x8*>(ap1 + k + 32 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = T::mfma16(fw[u], fa0[u], acc0);
      acc1 = T::mfma16(fw[u], fa1[u], acc1);
    }
  }
  for (; k < Kc; k += 32) {
    const s16x8 fw = *reinterpret_cast<const s16x8*>(wp + k);
    const s16x8 fa0 = *reinterpret_cast<const s16x8*>(ap0 + k);
    const s16x8 fa1 = *reinterpret_cast<const s16x8*>(ap1 + k);
    acc0 = T::mfma16(fw, fa0, acc0);
    acc1 = T::mfma16(fw, fa1, acc1);
  }
  // lane holds C[m = m0 + mt*16 + r][n = n0 + 4c + j]
  const int n = n0 + 4 * c;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + mt * 16 + r;
    if (m >= a.M) continue;
    const f32x4 acc = mt ? acc1 : acc0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n + j >= a.N) continue;
      float v = acc[j];
      if (MODE == SK_PARTIAL) {
        reinterpret_cast<float*>(a.out)[((int64_t)split * a.M + m) * a.N + n + j] = v;
        continue;
      }
      if (a.bias) v += a.bias[n + j];
      if (a.act == MIA_ACT_GELU) v = gelu_erf(v);
      if (MODE == SK_OUTF32) reinterpret_cast<float*>(a.out)[(int64_t)m * a.ldo + n + j] = v;
      else if (MODE == SK_OUT16) reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + n + j] = T::from_f32(v);
      else {  // SK_QKV: [0,D) -> q, [D,2D) -> self K cache, [2D,3D) -> self V cache at position pos
        const int nn = n + j;
        if (nn < a.D) reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + nn] = T::from_f32(v);
        else {
          const int hd = (nn - a.D) % a.D, h = hd >> 6, d = hd & 63;
          uint16_t* cache = nn < 2 * a.D ? a.cache_k : a.cache_v;
          cache[(((int64_t)m * a.H + h) * a.n_ctx + a.st->pos) * 64 + d] = T::from_f32(v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Single-query attention against a head-major K/V cache: one workgroup per (head, clip).
// 8 lanes share a key (16 B each, fully coalesced 1 KB per wave instruction); fp32 softmax.
// n_keys = st->pos + 1 (self attention) or the constant T (cross attention).
// ------------------------------------------------------------------------------------------------
constexpr int DEC_MAX_KEYS = 1536;

template <typename T>
__global__ __launch_bounds__(256) void dec_attention(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                     const uint16_t* __restrict__ vc, uint16_t* __restrict__ out,
                                                     const DecState* __restrict__ st, int fixed_keys, int cap_keys, int H,
                                                     float scale) {
  __shared__ float sc[DEC_MAX_KEYS];
  __shared__ float red[4][64];
  __shared__ float red2[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int D = H * 64;
  const int nk = fixed_keys > 0 ? fixed_keys : st->pos + 1;
  const int c = lane & 7, g = lane >> 3;
  const uint16_t* kb = kc + ((int64_t)b * H + h) * cap_keys * 64;
  const uint16_t* vb = vc + ((int64_t)b * H + h) * cap_keys * 64;
  float qf[8];
  {
    const s16x8 qv = *reinterpret_cast<const s16x8*>(q + (int64_t)b * D + h * 64 + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = T::to_f32((uint16_t)qv[j]);
  }
  // ---- scores
  for (int k0 = wave * 8; k0 < nk; k0 += 32) {
    const int key = k0 + g;
    float dot = 0.f;
    if (key < nk) {
      const s16x8 kv = *reinterpret_cast<const s16x8*>(kb + (int64_t)key * 64 + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += qf[j] * T::to_f32((uint16_t)kv[j]);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    if (c == 0 && key < nk) sc[key] = dot * scale;
  }
  __syncthreads();
  // ---- softmax over sc[0..nk)
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
  // ---- out = P V
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k0 = wave * 8; k0 < nk; k0 += 32) {
    const int key = k0 + g;
    if (key < nk) {
      const float p = sc[key];
      const s16x8 vv = *reinterpret_cast<const s16x8*>(vb + (int64_t)key * 64 + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += p * T::to_f32((uint16_t)vv[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] += __shfl_xor(acc[j], 8, 64);
    acc[j] += __shfl_xor(acc[j], 16, 64);
    acc[j] += __shfl_xor(acc[j], 32, 64);
  }
  if (g == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][c * 8 + j] = acc[j];
  }
  __syncthreads();
  if (tid < 64) {
    const float o = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) / sum;
    out[(int64_t)b * D + h * 64 + tid] = T::from_f32(o);
  }
}

// ------------------------------------------------------------------------------------------------
// Decode head: logit rules + argmax + log-prob bookkeeping (WhisperDecoding.swift:158-169,186-358), one workgroup
// per clip, all reductions in fp32.  Also advances nothing: dec_advance bumps the position afterwards.
// ------------------------------------------------------------------------------------------------
struct HeadBufs {
  const float* logits;          // [B][V]
  int32_t* tokens;              // [B][n_ctx]
  int32_t* n_gen;               // [B]
  int32_t* finished;            // [B]
  int32_t* last_ts;             // [B] last generated token value > timestamp_begin (0 = none)
  float* sum_logprob;           // [B]
  int32_t* n_logprob;           // [B]
  float* no_speech;             // [B]
  const uint32_t* suppress;     // [2][nw]
  const DecState* st;
};

__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

__global__ __launch_bounds__(1024) void dec_head(HeadBufs hb, DecodeParams p) {
  __shared__ float sh[16];
  __shared__ int shi[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int V = p.V;
  const float* lg = hb.logits + (int64_t)b * V;
  const int pos = hb.st->pos;
  const int cur_len = pos + 1;
  int32_t* toks = hb.tokens + (int64_t)b * p.n_ctx;

  // raw log-sum-exp of the logits (used by the no-speech probe and the timestamp heuristic)
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, lg[i]);
  mx = block_max(mx, sh);
  float se = 0.f;
  for (int i = tid; i < V; i += 1024) se += __expf(lg[i] - mx);
  se = block_sum(se, sh);
  const float lse = mx + __logf(se);

  if (pos == p.sot_index && tid == 0) hb.no_speech[b] = __expf(lg[p.no_speech] - lse);   // softmax(logits[sot])[no_speech]
  if (cur_len < p.n_initial) return;                  // next token is forced by the initial sequence
  if (hb.finished[b]) {
    if (tid == 0 && cur_len < p.n_ctx) toks[cur_len] = p.eot;
    return;
  }
  const int num_gen = cur_len - p.n_initial;          // == loop iteration of the reference
  const int tsb = p.timestamp_begin;

  // ---- rule state (WhisperDecoding.swift:221-292)
  bool sup_ts_all = false, sup_text_below_eot = false, sup_below_tsb = false;
  int ts_floor = 0;                                   // suppress tsb <= idx < ts_floor
  int max_first = V;                                  // suppress idx > max_first (first token only)
  if (p.timestamps) {
    const int last = toks[cur_len - 1];
    const bool last_was_ts = num_gen >= 1 && last >= tsb;
    const bool penult_was_ts = num_gen < 2 || toks[cur_len - 2] >= tsb;
    if (last_was_ts) { if (penult_was_ts) sup_ts_all = true; else sup_text_below_eot = true; }
    const int lt = hb.last_ts[b];
    if (lt > 0) ts_floor = penult_was_ts ? lt + 1 : lt;
    if (num_gen == 0) {
      sup_below_tsb = true;
      const int last_allowed = tsb + p.max_initial_ts;
      if (last_allowed < V) max_first = last_allowed;
    }
    if (num_gen > 0) {
      // heuristic on RAW logits (:299-322): logsumexp of timestamp log-probs vs max text log-prob
      float tmax = -INFINITY, xmax = -INFINITY;
      for (int i = tid; i < V; i += 1024) {
        const float lp = lg[i] - lse;
        if (i >= tsb) tmax = fmaxf(tmax, lp); else xmax = fmaxf(xmax, lp);
      }
      tmax = block_max(tmax, sh);
      xmax = block_max(xmax, sh);
      float ts = 0.f;
      for (int i = tsb + tid; i < V; i += 1024) ts += __expf((lg[i] - lse) - tmax);
      ts = block_sum(ts, sh);
      const float ts_lse = tmax + __logf(ts);
      if (ts_lse > xmax) sup_below_tsb = true;
    }
  }
  const int nw = (V + 31) / 32;
  const uint32_t* bits = hb.suppress + (num_gen == 0 ? nw : 0);
  auto masked = [&](int i) -> bool {
    if ((bits[i >> 5] >> (i & 31)) & 1u) return true;
    if (p.timestamps) {
      if (i == p.no_timestamps) return true;
      if (sup_ts_all && i >= tsb) return true;
      if (sup_text_below_eot && i < p.eot) return true;
      if (i >= tsb && i < ts_floor) return true;
      if (sup_below_tsb && i < tsb) return true;
      if (i > max_first) return true;
    }
    return false;
  };
  // ---- argmax over the filtered logits (lowest index wins ties) + their log-sum-exp
  float best = -INFINITY; int besti = 0x7fffffff;
  float fmx = -INFINITY;
  for (int i = tid; i < V; i += 1024) {
    if (masked(i)) continue;
    const float v = lg[i];
    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
  }
  fmx = block_max(best, sh);
  // index of the max: smallest index among threads holding the max value
  int cand = (best == fmx) ? besti : 0x7fffffff;
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) shi[tid >> 6] = cand;
  __syncthreads();
  int next = shi[0];
  for (int i = 1; i < 16; ++i) next = min(next, shi[i]);
  // Everything masked: reachable in the reference when the raw-logit timestamp heuristic (:299-322) fires right
  // after a timestamp pair rule; MLX argMax of an all -inf vector is index 0 and log(softmax) is NaN. Mirror that.
  const bool all_masked = next == 0x7fffffff;
  if (all_masked) next = 0;
  float fse = 0.f;
  for (int i = tid; i < V; i += 1024) if (!masked(i)) fse += __expf(lg[i] - fmx);
  fse = block_sum(fse, sh);
  if (tid == 0) {
    if (next != p.eot) {                              // EOT excluded from avg_logprob (:345-350)
      hb.sum_logprob[b] += all_masked ? __int_as_float(0x7fc00000) : (lg[next] - fmx) - __logf(fse);
      hb.n_logprob[b] += 1;
    }
    toks[cur_len] = next;
    hb.n_gen[b] = num_gen + 1;
    if (next > tsb) hb.last_ts[b] = next;             // strict '>' (:254-256)
    int cap = p.max_tokens - p.n_initial;
    if (p.max_new_tokens > 0 && p.max_new_tokens < cap) cap = p.max_new_tokens;
    if (next == p.eot || num_gen + 1 >= cap) hb.finished[b] = 1;
  }
}

__global__ void dec_advance(DecState* st) { st->pos += 1; }

// compact outputs: generated tokens with EOT (and anything after) stripped; avg_logprob
__global__ void dec_finalize(const int32_t* __restrict__ tokens, const int32_t* __restrict__ n_gen,
                             const float* __restrict__ sum_lp, const int32_t* __restrict__ n_lp, int32_t* __restrict__ out_tokens,
                             int32_t* __restrict__ out_n, float* __restrict__ out_avg, DecodeParams p) {
  const int b = blockIdx.x;
  const int32_t* t = tokens + (int64_t)b * p.n_ctx + p.n_initial;
  __shared__ int n_keep;
  if (threadIdx.x == 0) {
    int n = n_gen[b];
    for (int i = 0; i < n; ++i) if (t[i] == p.eot) { n = i; break; }
    n_keep = n;
    out_n[b] = n;
    out_avg[b] = n_lp[b] > 0 ? sum_lp[b] / (float)n_lp[b] : 0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.max_tokens; i += blockDim.x) out_tokens[(int64_t)b * p.max_tokens + i] = i < n_keep ? t[i] : 0;
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
int dec_launch_embed_ln(mia_whisper* w, const LNW& ln, hipStream_t s) {
  const int D = w->dims.n_text_state;
  if (w->dtype == MIA_F16)
    hipLaunchKernelGGL(dec_embed_ln<F16>, dim3(w->cur_B), dim3(64), 0, s, w->tokens, (const uint16_t*)w->tok_emb, w->dec_pos, ln.g, ln.b, w->dx, (uint16_t*)w->dh, w->state, D, w->dims.n_text_ctx);
  else
    hipLaunchKernelGGL(dec_embed_ln<BF16>, dim3(w->cur_B), dim3(64), 0, s, w->tokens, (const uint16_t*)w->tok_emb, w->dec_pos, ln.g, ln.b, w->dx, (uint16_t*)w->dh, w->state, D, w->dims.n_text_ctx);
  return 0;
}

int dec_launch_reduce_ln(mia_whisper* w, int S, const float* bias, const LNW& ln, hipStream_t s) {
  const int D = w->dims.n_text_state;
  if (w->dtype == MIA_F16)
    hipLaunchKernelGGL(dec_reduce_ln<F16>, dim3(w->cur_B), dim3(64), 0, s, w->partial, S, w->cur_B, bias, ln.g, ln.b, w->dx, (uint16_t*)w->dh, D);
  else
    hipLaunchKernelGGL(dec_reduce_ln<BF16>, dim3(w->cur_B), dim3(64), 0, s, w->partial, S, w->cur_B, bias, ln.g, ln.b, w->dx, (uint16_t*)w->dh, D);
  return 0;
}

template <typename T>
static void skinny_launch_t(const SkinnyArgs& a, int mode, hipStream_t s) {
  dim3 grid((a.N + 15) / 16, a.S, (a.M + 31) / 32), block(64);
  switch (mode) {
    case SK_OUT16: hipLaunchKernelGGL((dec_skinny_gemm<T, SK_OUT16>), grid, block, 0, s, a); break;
    case SK_OUTF32: hipLaunchKernelGGL((dec_skinny_gemm<T, SK_OUTF32>), grid, block, 0, s, a); break;
    case SK_PARTIAL: hipLaunchKernelGGL((dec_skinny_gemm<T, SK_PARTIAL>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((dec_skinny_gemm<T, SK_QKV>), grid, block, 0, s, a); break;
  }
}

int dec_launch_skinny(mia_whisper* w, const SkinnyArgs& a, int mode, hipStream_t s) {
  if (a.K % (32 * a.S) != 0 || a.lda % 8 != 0) return -1;
  if (w->dtype == MIA_F16) skinny_launch_t<F16>(a, mode, s); else skinny_launch_t<BF16>(a, mode, s);
  return 0;
}

int dec_launch_attention(mia_whisper* w, const void* q, const void* kc, const void* vc, void* out, int fixed_keys, int cap_keys,
                         hipStream_t s) {
  if (cap_keys > DEC_MAX_KEYS) return -1;
  dim3 grid(w->dims.n_text_head, w->cur_B), block(256);
  if (w->dtype == MIA_F16)
    hipLaunchKernelGGL(dec_attention<F16>, grid, block, 0, s, (const uint16_t*)q, (const uint16_t*)kc, (const uint16_t*)vc, (uint16_t*)out, w->state, fixed_keys, cap_keys, w->dims.n_text_head, 0.125f);
  else
    hipLaunchKernelGGL(dec_attention<BF16>, grid, block, 0, s, (const uint16_t*)q, (const uint16_t*)kc, (const uint16_t*)vc, (uint16_t*)out, w->state, fixed_keys, cap_keys, w->dims.n_text_head, 0.125f);
  return 0;
}

int dec_launch_head(mia_whisper* w, int32_t* last_ts, const DecodeParams& p, hipStream_t s) {
  HeadBufs hb{w->logits, w->tokens, w->n_gen, w->finished, last_ts, w->sum_logprob, w->n_logprob, w->no_speech, w->suppress_bits, w->state};
  hipLaunchKernelGGL(dec_head, dim3(w->cur_B), dim3(1024), 0, s, hb, p);
  hipLaunchKernelGGL(dec_advance, dim3(1), dim3(1), 0, s, w->state);
  return 0;
}

int dec_launch_finalize(mia_whisper* w, int32_t* out_n, const DecodeParams& p, hipStream_t s) {
  hipLaunchKernelGGL(dec_finalize, dim3(w->cur_B), dim3(256), 0, s, w->tokens, w->n_gen, w->sum_logprob, w->n_logprob, w->out_tokens, out_n, w->out_avg, p);
  return 0;
}
