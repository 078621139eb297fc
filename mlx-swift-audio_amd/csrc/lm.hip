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
#include "gemm.h"

struct LmState { int pos; int n_hist; int finished; int n_gen; int n_embeds; int n_out; int u_cursor; int n_prompt; int min_len; int max_len; };   // min/max_len: RAS loop, per sequence

struct RasParams { float top_p; int top_k; int win; float tau; int eos; int min_len; int max_len; int n_uniforms; };

// MLX-affine 4- / 8-bit copy of one fused matrix in MFMA fragment order (decode_kernels.hip: skinny_gemm_qi); null = use the 16-bit weights
struct Q4W { uint32_t* wfrag = nullptr; float* stfrag = nullptr; };

struct LmLayer {
  Q4W q_qkv, q_o, q_gu, q_down;
  float* in_norm = nullptr; float* post_norm = nullptr;
  void* wqkv = nullptr; float* bqkv = nullptr;   // [(Hq+2Hkv)*dh][hidden]
  void* wo = nullptr;                            // [hidden][Hq*dh]
  void* wgu = nullptr;                           // [2*inter][hidden], rows interleaved gate/up
  void* wdown = nullptr;                         // [hidden][inter]
  // the same four in MFMA-fragment order (decode.h) for the decode step's skinny GEMMs: one contiguous 1 KB per wave load instead of
  // 16 rows x 64 B (tools/micro/skinny_probe.hip, Orpheus-3B shapes, one sequence: 48.3 -> 43.7 us per layer); the batched prompt
  // pass (gemm.hip) keeps reading the row-major copies
  void* wqkv_f = nullptr; void* wo_f = nullptr; void* wgu_f = nullptr; void* wdown_f = nullptr;
};

struct mia_lm {
  mia_ctx* ctx = nullptr;
  mia_lm_config cfg{};
  int dtype = MIA_BF16;
  std::vector<void*> allocs;
  void* embed = nullptr;        // 16-bit [V][hidden]
  void* lm_head = nullptr;      // 16-bit [V][hidden] (== embed when tied)
  void* lm_head_f = nullptr;    // lm_head in MFMA-fragment order (decode step)
  Q4W q_head;                   // 4-bit copy of lm_head (mia_lm_attach_q4)
  int q_bits = 0;               // 4 | 8 once packed weights are attached (mia_lm_attach_quantized), 0 = none
  bool q4 = false;              // the step GEMVs stream the packed weights
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
  float* ss = nullptr;          // [2][hidden / 16][B]: per-tile partial sums of squares of the residual stream (SK_RESID producers)
  int32_t* tokens = nullptr;    // [max_ctx] full sequence
  int32_t* hist = nullptr;      // [64] repetition window (ring, oldest first)
  float* uniforms = nullptr;    // [max_ctx]
  void* smx = nullptr;            // SmxWs[B]: sampler scratch, one per sequence (slice records + radix slabs)
  LmState* state = nullptr;
  hipGraphExec_t graph = nullptr;       // one decode step (forward / top-p sampler / RAS sampler), re-captured when its sampler arguments change
  int graph_mode = -1;                  // 0 forward, 1 top-p, 2 RAS
  int debug_flags = 0;                  // test hook (mia_lm_set_debug): bit 0 = no hipGraph, bit 1 = no batched prompt pass
  mia_lm_sampler graph_sampler{};
  RasParams graph_ras{};
  int S_qkv = 1, S_o = 1, S_down = 1;
  // batched prompt pass (lm_prefill): row buffers for one chunk of PF_ROWS positions, allocated on first use
  char* pf_buf = nullptr;
  // sequences decoded side by side (mia_lm_set_batch): every state buffer above holds B_cap rows / caches; the single-sequence entry
  // points use row 0
  int B_cap = 1;
  int graph_nb = 0;
  std::vector<void*> state_allocs;
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
  // the norm weights do not depend on the reduction: fetch them before it, so their L2 round trip overlaps the two barriers
  f32x4 g[LM_NV];
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) g[i] = tid + 256 * i < nv ? *reinterpret_cast<const f32x4*>(w + 4 * (tid + 256 * i)) : (f32x4){0.f, 0.f, 0.f, 0.f};
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) if (tid + 256 * i < nv) q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
  const float rstd = rsqrtf(blk256_sum(q, sh) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) {
    const int c = tid + 256 * i;
    if (c >= nv) continue;
    *reinterpret_cast<u32x2*>(h + 4 * c) = (u32x2){pack2<T>(v[i][0] * rstd * g[i][0], v[i][1] * rstd * g[i][1]), pack2<T>(v[i][2] * rstd * g[i][2], v[i][3] * rstd * g[i][3])};
  }
}

// x = E[token[pos]] (or a caller-provided embedding row);  h = RMSNorm(x) * w
// One workgroup per row.  Step graph (rowmap == nullptr): row b is SEQUENCE b at its own st[b].pos (tokens / embedding rows / state strided
// per sequence); batched prompt pass: row r is position rowmap[r].y of sequence rowmap[r].x -- rows of several prompts share a pass.
template <typename T>
__global__ __launch_bounds__(256) void lm_embed_norm(const int32_t* __restrict__ tokens, const uint16_t* __restrict__ emb, const uint16_t* __restrict__ gen_emb,
                                                     const float* __restrict__ embeds, const float* __restrict__ w,
                                                     float* __restrict__ x, uint16_t* __restrict__ h, const LmState* __restrict__ st, int D, float eps,
                                                     const int2* __restrict__ rowmap, int max_ctx, int emb_rows, int gen_rows) {
  __shared__ float sh[4];
  const int tid = threadIdx.x, nv = D >> 2;
  const int seq = rowmap ? rowmap[blockIdx.x].x : (int)blockIdx.x;         // prompt pass: row -> (sequence, position)
  st += seq; tokens += (int64_t)seq * max_ctx; embeds += (int64_t)seq * max_ctx * D;
  const int pos = rowmap ? rowmap[blockIdx.x].y : st->pos;
  x += (int64_t)blockIdx.x * D; h += (int64_t)blockIdx.x * D;
  const bool from_rows = pos < st->n_embeds;               // prompt given as embedding rows
  const bool use_gen = st->n_embeds > 0 && gen_emb;
  int tok = from_rows ? 0 : tokens[pos];
  const int rows = use_gen ? gen_rows : emb_rows;
  tok = tok < 0 ? 0 : (tok < rows ? tok : rows - 1);       // an id outside the table (a corrupted sampler output) must never become a wild address
  const uint16_t* e = (use_gen ? gen_emb : emb) + (int64_t)tok * D;
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

// x += sum_s partial[s];  h = RMSNorm(x) * w        (one workgroup per row; the batched prompt pass runs it with S = 0)
template <typename T>
__global__ __launch_bounds__(256) void lm_reduce_norm(const float* __restrict__ partial, int S, const float* __restrict__ w, float* __restrict__ x,
                                                      uint16_t* __restrict__ h, int D, float eps, int B) {
  __shared__ float sh[4];
  const int tid = threadIdx.x, nv = D >> 2;
  x += (int64_t)blockIdx.x * D; h += (int64_t)blockIdx.x * D;
  f32x4 v[LM_NV];
#pragma unroll
  for (int i = 0; i < LM_NV; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      f32x4 a = *reinterpret_cast<const f32x4*>(x + 4 * c);
      // the slices were written by other XCDs (L2 misses): issue all loads before the first add.  S <= 8; slot k >= S re-reads
      // slice S-1 and is discarded, so the loads are unconditional and the sum keeps its fixed order.
      f32x4 p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)                        // slices are [S][B][D]: row blockIdx.x of each
        p[k] = S > 0 ? *reinterpret_cast<const f32x4*>(partial + ((int64_t)(k < S ? k : S - 1) * B + blockIdx.x) * D + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < S) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += p[k][j];
        }
      v[i] = a;
      *reinterpret_cast<f32x4*>(x + 4 * c) = a;
    } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  rms_store<T>(v, nv, D, eps, w, h, sh);
}

// q|k|v = sum_s partial[s] + bias; RoPE (split-half pairs (i, i+dh/2), angle = pos * inv_freq[i]) on q and k; q -> qout,
// k, v -> cache[kv head][pos][:]          one thread per rotation pair / per v element; blockIdx.y = row of the batched prompt pass
template <typename T>
__global__ __launch_bounds__(256) void lm_rope_cache(const float* __restrict__ part, int S, const float* __restrict__ bias, const float* __restrict__ inv_freq,
                                                     uint16_t* __restrict__ qout, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                     const int2* __restrict__ rowmap, int Hq, int Hkv, int dh, int max_ctx, int64_t seq_stride) {
  const int Nq = Hq * dh, Nk = Hkv * dh, N = Nq + 2 * Nk;
  const int half = dh >> 1;
  const int pos = rowmap[blockIdx.y].y;
  kc += (int64_t)rowmap[blockIdx.y].x * seq_stride; vc += (int64_t)rowmap[blockIdx.y].x * seq_stride;
  part += (int64_t)blockIdx.y * N; qout += (int64_t)blockIdx.y * Nq;      // prompt pass: S == 1, one GEMM output row per position
  const int n_pairs = (Hq + Hkv) * half;
  const int e = blockIdx.x * 256 + threadIdx.x;
  auto val = [&](int n) {        // S <= 4 slices, loads issued together (see lm_reduce_norm)
    float p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = part[(int64_t)(k < S ? k : S - 1) * N + n];
    float a = bias ? bias[n] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < S) a += p[k];
    return a;
  };
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

// grouped-query single-token attention: one workgroup per (query head, row); DH/8 lanes share a key.  Row r of the batched
// prompt pass sits at position pos0 + r and sees keys [0, pos0 + r] -- causal by construction, same arithmetic as a decode step.
// 16 waves per workgroup: a decode step has only n_heads workgroups, so the key loop's memory latency is hidden by waves of
// the same workgroup (4 waves: +12 us per layer per 300 keys on Orpheus-3B; 16 waves: a quarter of that).
constexpr int ATT_NW = 16;
// FUSED (the decode step): the workgroup first finishes its own q head and its K/V head's new row from the split-K slices of the
// q|k|v GEMM (sum + bias + RoPE, exactly lm_rope_cache's arithmetic and 16-bit rounding), keeps them in LDS, and the first query
// head of each group writes the K/V row to the cache -- one kernel less per layer; the new row is used from LDS because the
// writer may be another workgroup.
template <typename T, int DH, bool FUSED>
__global__ __launch_bounds__(64 * ATT_NW) void lm_attention(const uint16_t* __restrict__ q, uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                    uint16_t* __restrict__ out, const LmState* __restrict__ st, int Hq, int Hkv, int max_ctx, float scale, int pos0,
                                                    const float* __restrict__ part, int S, const float* __restrict__ bias, const float* __restrict__ inv_freq,
                                                    int B, int64_t seq_stride, const int2* __restrict__ rowmap) {
  extern __shared__ float sc[];            // [max_ctx] scores, then red[ATT_NW][DH] + red2[2 * ATT_NW] (+ FUSED: q, k, v rows [3][DH])
  const int seq = FUSED ? (int)blockIdx.y : rowmap[blockIdx.y].x;          // FUSED: row = sequence; prompt pass: row -> (sequence, position)
  st += seq; kc += (int64_t)seq * seq_stride; vc += (int64_t)seq * seq_stride;
  constexpr int LPK = DH / 8;              // lanes per key
  constexpr int KPW = 64 / LPK;            // keys per wave instruction
  float* red = sc + max_ctx;
  float* red2 = red + ATT_NW * DH;
  float* qs = red2 + 2 * ATT_NW;           // FUSED only
  float* kn = qs + DH;
  float* vn = kn + DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, kvh = h / (Hq / Hkv);
  const int nk = (FUSED ? st->pos : rowmap[blockIdx.y].y) + 1;
  const int nkc = FUSED ? nk - 1 : nk;     // keys read from the cache
  q += (int64_t)blockIdx.y * Hq * DH; out += (int64_t)blockIdx.y * Hq * DH;
  const int c = lane % LPK, g = lane / LPK;
  const uint16_t* kb = kc + (int64_t)kvh * max_ctx * DH;
  const uint16_t* vb = vc + (int64_t)kvh * max_ctx * DH;
  if (FUSED) {
    constexpr int half = DH / 2;
    const int Nq = Hq * DH, Nk = Hkv * DH, N = Nq + 2 * Nk, pos = nk - 1;
    const bool writer = h % (Hq / Hkv) == 0;
    auto val = [&](int n) {                // S <= 4 slices, loads issued together
      float pv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pv[k] = part[((int64_t)(k < S ? k : S - 1) * B + blockIdx.y) * N + n];   // slices [S][B][N]
      float a = bias ? bias[n] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < S) a += pv[k];
      return a;
    };
    if (tid < DH) {                        // rotation pairs (i, i + half): the q head, then the k head
      const bool is_k = tid >= half;
      const int i = is_k ? tid - half : tid;
      const int base = is_k ? Nq + kvh * DH : h * DH;
      const float x0 = val(base + i), x1 = val(base + i + half);
      float sn, cs;
      sincosf((float)pos * inv_freq[i], &sn, &cs);
      const uint16_t r0 = T::from_f32(x0 * cs - x1 * sn), r1 = T::from_f32(x1 * cs + x0 * sn);
      float* dst = is_k ? kn : qs;
      dst[i] = T::to_f32(r0); dst[i + half] = T::to_f32(r1);
      if (is_k && writer) { uint16_t* k = kc + ((int64_t)kvh * max_ctx + pos) * DH; k[i] = r0; k[i + half] = r1; }
    } else if (tid < 2 * DH) {
      const int d = tid - DH;
      const uint16_t r = T::from_f32(val(Nq + Nk + kvh * DH + d));
      vn[d] = T::to_f32(r);
      if (writer) vc[((int64_t)kvh * max_ctx + pos) * DH + d] = r;
    }
    __syncthreads();
  }
  float qf[8];
  if (FUSED) {
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = qs[c * 8 + j];
  } else {
    const s16x8 qv = *reinterpret_cast<const s16x8*>(q + h * DH + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = T::to_f32((uint16_t)qv[j]);
  }
  for (int k0 = wave * KPW * 4; k0 < nkc; k0 += ATT_NW * KPW * 4) {
    s16x8 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int key = k0 + KPW * u + g; key = key < nkc ? key : nkc - 1; kv[u] = *reinterpret_cast<const s16x8*>(kb + (int64_t)key * DH + c * 8); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += qf[j] * T::to_f32((uint16_t)kv[u][j]);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) dot += __shfl_xor(dot, o, 64);
      const int key = k0 + KPW * u + g;
      if (c == 0 && key < nkc) sc[key] = dot * scale;
    }
  }
  if (FUSED && wave == 0) {                // the new key, from LDS (same lane split and summation order as a cached key)
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += qf[j] * kn[c * 8 + j];
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) dot += __shfl_xor(dot, o, 64);
    if (lane == 0) sc[nk - 1] = dot * scale;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int i = tid; i < nk; i += 64 * ATT_NW) m = fmaxf(m, sc[i]);
  m = wave_max(m);
  if (lane == 0) red2[wave] = m;
  __syncthreads();
  m = red2[0];
#pragma unroll
  for (int i = 1; i < ATT_NW; ++i) m = fmaxf(m, red2[i]);
  float sum = 0.f;
  for (int i = tid; i < nk; i += 64 * ATT_NW) { const float p = __expf(sc[i] - m); sc[i] = p; sum += p; }
  sum = wave_sum(sum);
  if (lane == 0) red2[ATT_NW + wave] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int i = 0; i < ATT_NW; ++i) sum += red2[ATT_NW + i];
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k0 = wave * KPW * 4; k0 < nkc; k0 += ATT_NW * KPW * 4) {
    s16x8 vv[4]; float pw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + KPW * u + g; const int k2 = key < nkc ? key : nkc - 1;
      vv[u] = *reinterpret_cast<const s16x8*>(vb + (int64_t)k2 * DH + c * 8);
      pw[u] = key < nkc ? sc[k2] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pw[u] * T::to_f32((uint16_t)vv[u][j]);
  }
  if (FUSED && wave == 0 && g == 0) {      // the new row's contribution
    const float pw = sc[nk - 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += pw * vn[c * 8 + j];
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
  if (tid < DH) {
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_NW; ++i) o += red[i * DH + tid];
    out[h * DH + tid] = T::from_f32(o / sum);
  }
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

template <typename V>
__device__ __forceinline__ V wave_incl_scan(V v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const V t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
  return v;
}

// ---- the top-p sampler, split over the vocabulary: SMX_G workgroups per sequence, one kernel per dependent phase ----
// One workgroup walking V = 156 940 logits six times took 82-91 us per token (46 GB/s: what one CU streams), 4-7 % of an Orpheus-3B
// step.  Here each of 32 workgroups owns one contiguous slice of the vocabulary (4 905 tokens at that V), holds it in registers
// (24 loads in flight per thread: one memory round trip per kernel) and the phases that need a vocabulary-wide result are separate
// kernels of the step graph (a kernel boundary inside a hipGraph costs 1.5 us, a grid barrier 4-5):
//   smx_max   repetition penalty on the slice's tokens; slice maximum of z / T
//   smx_exp   p = exp(z / T - max) written in place; slice sum; radix level 0 (sign | exponent) histogram of the slice
//   smx_level<1..3>  walk the previous level's merged histogram down to the bin where the descending cumulative sum crosses
//             top_p * total, then histogram the next 8 / 8 / 7 mantissa bits of the slice's members of that bin (level 3 also sums the
//             slice's probabilities ABOVE the bin: they are kept whatever the last 7 bits turn out to be)
//   smx_draw  one workgroup: walks the last level -> the exact bit pattern of the smallest kept probability + how many of its ties are
//             kept; every slice's kept sum and tie count from its level-3 slab; inverse-CDF draw in index order with the caller's
//             uniform (slice, then chunk, then token) + the sequence's bookkeeping
// Histograms: a bin is ONE u64, (count << 44) | sum of the bits below the level's digit -- every p of a bin shares the bits above, so
// the bin's sum is exactly count * base + ulp * sum(low): integer LDS atomics (tools/micro/lds_atomic_rate.hip: ds_add_f32 0.8
// lane-ops/ns, ds_add_u64 25), order-independent and exact, so the slices' histograms add up to the same u64 in any order.  A slice
// writes its 256-bin slab to global memory, the next kernel's workgroups each add the 32 slabs (every workgroup repeats the same walk
// and reaches the same bin; workgroup 0 records the level's result for the kernels after it -- in a slot of its own, since the other
// workgroups of the same launch are still reading the previous level's).  Ties at the threshold are kept lowest index first.
constexpr int SMX_G = 32;                              // vocabulary slices = workgroups per sequence
constexpr int SMX_NT = 256;                            // threads per workgroup (4 waves); also the widest level's bin count
constexpr int SMX_NW = SMX_NT / 64;
constexpr int SMX_U = 24;                              // loads in flight per thread: one batch covers a slice of 6 144 tokens (V <= 196 608)
constexpr int SMX_LV = 4;
constexpr int SMX_W[SMX_LV] = {9, 8, 8, 7};            // digit widths, top down: sign|exponent, then 23 mantissa bits
constexpr int SMX_SH[SMX_LV] = {23, 15, 7, 0};
constexpr int SMX_NB[SMX_LV] = {128, 256, 256, 128};   // (0 <= p <= 1: the top digit is <= 127)
constexpr int SMX_CP = 8, SMX_STRIDE = 257;            // LDS copies per lane group (a wave whose lanes all hit one bin must not serialise)
constexpr int SMX_CNT_SHIFT = 44;                      // V < 2^20 tokens, <= 23-bit `low`: 43 bits of sum

struct SmxLevel { double cum_above; double target; unsigned prefix, mask; };
struct SmxWs {                                         // per sequence
  float pmax[SMX_G], psum[SMX_G], pabove[SMX_G];
  SmxLevel lv[SMX_LV];                                 // lv[l]: the state after walking level l (l = 0 .. 2)
  unsigned long long slab[SMX_LV][SMX_G][SMX_NT];
};

struct SmxArgs { float* logits; int V; int32_t* tokens; int32_t* hist; const float* uniforms; LmState* st; SmxWs* ws; mia_lm_sampler sp; int n_prompt; int max_ctx; };
struct SmxCtx { float* P; int32_t* tokens; int32_t* hist; const float* uniforms; LmState* st; SmxWs* ws; int lo, hi, sl; };

// this workgroup's sequence (blockIdx.y) and slice (blockIdx.x): pointers and bounds only, no memory access
__device__ __forceinline__ SmxCtx smx_ctx(const SmxArgs& a) {
  SmxCtx c;
  const int b = blockIdx.y;
  c.st = a.st + b; c.P = a.logits + (int64_t)b * a.V; c.tokens = a.tokens + (int64_t)b * a.max_ctx; c.hist = a.hist + b * 64; c.uniforms = a.uniforms + (int64_t)b * a.max_ctx; c.ws = a.ws + b;
  c.sl = (a.V + SMX_G - 1) / SMX_G;
  c.lo = min(a.V, (int)blockIdx.x * c.sl); c.hi = min(a.V, c.lo + c.sl);
  return c;
}
// false while the prompt is still being consumed (or the sequence is done): nothing to draw.  (The step graph passes n_prompt = -1 and
// the state holds it: one graph serves every prompt length.)
__device__ __forceinline__ bool smx_drawing(const SmxArgs& a, const LmState& s) { return !(s.pos + 1 < (a.n_prompt < 0 ? s.n_prompt : a.n_prompt) || s.finished); }
__device__ __forceinline__ bool smx_use_top_p(const mia_lm_sampler& sp, int V) { return sp.top_p > 0.0f && sp.top_p < 1.0f && V > 1; }

// one batch of the slice, element u of thread t = token base + t + 256 u; the loads are unconditional (index clamped) so that they are
// all in flight before anything waits
__device__ __forceinline__ void smx_load(const float* __restrict__ P, int base, int hi, float (&v)[SMX_U]) {
  const int last = max(hi - 1, 0);
#pragma unroll
  for (int u = 0; u < SMX_U; ++u) v[u] = P[min(base + (int)threadIdx.x + SMX_NT * u, last)];
}
template <typename F>
__device__ __forceinline__ void smx_apply(const float (&v)[SMX_U], int base, int hi, F f) {
#pragma unroll
  for (int u = 0; u < SMX_U; ++u) { const int i = base + (int)threadIdx.x + SMX_NT * u; if (i < hi) f(i, v[u], u); }
}
// f over the whole slice; v0 = its first batch, already loaded
template <typename F>
__device__ __forceinline__ void smx_each(const float* __restrict__ P, int lo, int hi, const float (&v0)[SMX_U], F f) {
  smx_apply(v0, lo, hi, f);
  for (int base = lo + SMX_NT * SMX_U; base < hi; base += SMX_NT * SMX_U) { float v[SMX_U]; smx_load(P, base, hi, v); smx_apply(v, base, hi, f); }
}

template <typename V, int NW>
__device__ __forceinline__ V blk_excl_scan(V v, V* sh) {   // exclusive prefix over the workgroup in thread order; sh: NW words of scratch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const V incl = wave_incl_scan(v, lane);
  V excl = __shfl_up(incl, 1, 64);
  if (lane == 0) excl = (V)0;
  __syncthreads();
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  V base = (V)0;
#pragma unroll
  for (int i = 0; i < NW; ++i) if (i < wave) base += sh[i];
  return base + excl;
}
template <int NW>
__device__ __forceinline__ float blk_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += sh[i];
  return r;
}

// value of a bin record: count * (smallest member) + ulp * sum(low bits), exact in double
__device__ __forceinline__ double smx_bin_value(unsigned long long h, unsigned bits) {
  constexpr unsigned long long ONE = 1ull << SMX_CNT_SHIFT;
  const int e = max((int)((bits >> 23) & 0xffu), 1);
  const double ulp = __longlong_as_double((long long)(e - 150 + 1023) << 52);       // 2^(e - 150)
  return (double)(h >> SMX_CNT_SHIFT) * (double)__uint_as_float(bits) + (double)(h & (ONE - 1ull)) * ulp;
}

__device__ __forceinline__ void smx_hist_clear(unsigned long long* hb) {
  for (int i = threadIdx.x; i < SMX_CP * SMX_STRIDE; i += SMX_NT) hb[i] = 0ull;
}
template <int LVL>
__device__ __forceinline__ void smx_hist_put(unsigned long long* hb, float p, unsigned prefix, unsigned mask) {
  constexpr unsigned long long ONE = 1ull << SMX_CNT_SHIFT;
  constexpr int shf = SMX_SH[LVL], nb = SMX_NB[LVL];
  constexpr unsigned dmask = (1u << SMX_W[LVL]) - 1u, lmask = (1u << shf) - 1u;
  const unsigned b = __float_as_uint(p);
  if ((b & mask) != prefix) return;
  const int bin = min((int)((b >> shf) & dmask), nb - 1);
  atomicAdd(&hb[(threadIdx.x & (SMX_CP - 1)) * SMX_STRIDE + bin], ONE | (unsigned long long)(b & lmask));
}
// the slice's 256-bin slab to global memory (every bin written: no zeroing pass)
template <int LVL>
__device__ __forceinline__ void smx_hist_store(const SmxCtx& c, const unsigned long long* hb) {
  unsigned long long h = 0ull;
  if ((int)threadIdx.x < SMX_NB[LVL]) {
#pragma unroll
    for (int r = 0; r < SMX_CP; ++r) h += hb[r * SMX_STRIDE + threadIdx.x];
  }
  c.ws->slab[LVL][blockIdx.x][threadIdx.x] = h;
}

// Walk level LVL's merged histogram from the largest bin down until the cumulative sum crosses the target, as a workgroup-wide prefix
// sum: thread t owns descending position t (bin nb-1-t) and has the merged record h of its bin.  Returns the state after the level
// (identical in every thread of every workgroup of the launch); sel_bin = the bin it settled on.
template <int LVL>
__device__ __forceinline__ SmxLevel smx_walk(unsigned long long h, SmxLevel in, double* shd, int* s_i, double* s_d, int* sel_bin = nullptr) {
  constexpr int shf = SMX_SH[LVL], nb = SMX_NB[LVL];
  constexpr unsigned dmask = (1u << SMX_W[LVL]) - 1u;
  const int tid = threadIdx.x;
  if (tid == 0) { s_i[0] = 0x7fffffff; s_i[1] = -1; }
  const int bin = nb - 1 - tid;
  const bool occ = bin >= 0 && (h >> SMX_CNT_SHIFT) != 0ull;
  const double a = occ ? smx_bin_value(h, in.prefix | ((unsigned)bin << shf)) : 0.0;
  const double before = in.cum_above + blk_excl_scan<double, SMX_NW>(a, shd);      // (its barriers publish s_i's reset)
  if (occ && before + a > in.target) atomicMin(&s_i[0], tid);
  if (occ) atomicMax(&s_i[1], tid);
  __syncthreads();
  // never crossed (rounding corner): settle on the lowest occupied bin
  const int sel_r = s_i[0] != 0x7fffffff ? s_i[0] : max(s_i[1], 0);
  if (tid == sel_r) s_d[0] = before;
  __syncthreads();
  SmxLevel out = in;
  out.cum_above = s_d[0];
  out.prefix = in.prefix | (((unsigned)(nb - 1 - sel_r)) << shf);
  out.mask = in.mask | (dmask << shf);
  if (sel_bin) *sel_bin = nb - 1 - sel_r;
  __syncthreads();
  return out;
}
// thread t's merged record of bin nb-1-t: the 32 slices' slabs added up (all loads in flight at once)
template <int LVL>
__device__ __forceinline__ unsigned long long smx_merged(const SmxWs* ws) {
  const int bin = SMX_NB[LVL] - 1 - (int)threadIdx.x;
  const unsigned long long* sl = &ws->slab[LVL][0][max(bin, 0)];
  unsigned long long r[SMX_G];
#pragma unroll
  for (int g = 0; g < SMX_G; ++g) r[g] = sl[(size_t)g * SMX_NT];
  unsigned long long h = 0ull;
#pragma unroll
  for (int g = 0; g < SMX_G; ++g) h += r[g];
  return h;
}

__global__ __launch_bounds__(SMX_NT) void smx_max(SmxArgs a) {
  const SmxCtx c = smx_ctx(a);
  __shared__ float sh[SMX_NW];
  __shared__ int pen_tok[64];
  __shared__ float pen_val[64];
  __shared__ int n_pen;
  const int tid = threadIdx.x;
  float v0[SMX_U];
  smx_load(c.P, c.lo, c.hi, v0);
  const LmState s = *c.st;
  const int htok = tid < 64 ? c.hist[tid] : -1;
  if (tid == 0) n_pen = 0;
  if (!smx_drawing(a, s)) return;
  // 1. repetition penalty over the last `rep_window` generated tokens (gather all, then scatter: duplicates penalised once); a token
  //    lies in exactly one slice.  The slice is already in registers: the penalised values are patched in through a short LDS list
  __syncthreads();
  if (a.sp.rep_penalty != 1.0f && tid < s.n_hist && htok >= c.lo && htok < c.hi) {
    const float gth = c.P[htok];
    const float upd = gth < 0.f ? gth * a.sp.rep_penalty : gth / a.sp.rep_penalty;
    const int k = atomicAdd(&n_pen, 1);
    pen_tok[k] = htok; pen_val[k] = upd;
  }
  __syncthreads();
  const int np = n_pen;
  if (tid < np) c.P[pen_tok[tid]] = pen_val[tid];        // (duplicates write the same value)
  float mx = -INFINITY;
  smx_each(c.P, c.lo, c.hi, v0, [&](int i, float z, int) {
    for (int k = 0; k < np; ++k) z = pen_tok[k] == i ? pen_val[k] : z;
    mx = fmaxf(mx, z);
  });
  mx = wave_max(mx);
  if ((tid & 63) == 0) sh[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    float r = sh[0];
    for (int i = 1; i < SMX_NW; ++i) r = fmaxf(r, sh[i]);
    c.ws->pmax[blockIdx.x] = r * (1.0f / fmaxf(a.sp.temperature, 1e-6f));      // (T > 0: the maximum commutes with the scaling)
  }
}

__global__ __launch_bounds__(SMX_NT) void smx_exp(SmxArgs a) {
  const SmxCtx c = smx_ctx(a);
  __shared__ unsigned long long hb[SMX_CP * SMX_STRIDE];
  __shared__ float sh[SMX_NW];
  const int tid = threadIdx.x;
  float v0[SMX_U];
  smx_load(c.P, c.lo, c.hi, v0);
  float mx = c.ws->pmax[tid & (SMX_G - 1)];
  const LmState s = *c.st;
  if (!smx_drawing(a, s)) return;
  const bool use_top_p = smx_use_top_p(a.sp, a.V);
  smx_hist_clear(hb);
#pragma unroll
  for (int o = 1; o < SMX_G; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  const float inv_t = 1.0f / fmaxf(a.sp.temperature, 1e-6f);
  __syncthreads();
  // 2. the unnormalised probabilities p = exp(z / T - max) REPLACE the logits (the buffer is rewritten by the next step's head GEMM),
  //    so every later kernel reads the same p without another exp
  float tot = 0.f;
  smx_each(c.P, c.lo, c.hi, v0, [&](int i, float z, int) {
    const float p = __expf(z * inv_t - mx);
    c.P[i] = p; tot += p;
    if (use_top_p) smx_hist_put<0>(hb, p, 0u, 0u);
  });
  tot = blk_sum<SMX_NW>(tot, sh);                        // (its barriers also complete the histogram)
  if (tid == 0) c.ws->psum[blockIdx.x] = tot;
  if (use_top_p) smx_hist_store<0>(c, hb);
}

template <int LVL>      // 1 .. 3: walk level LVL-1, histogram level LVL
__global__ __launch_bounds__(SMX_NT) void smx_level(SmxArgs a) {
  const SmxCtx c = smx_ctx(a);
  if (!smx_use_top_p(a.sp, a.V)) return;
  __shared__ unsigned long long hb[SMX_CP * SMX_STRIDE];
  __shared__ double shd[SMX_NW];
  __shared__ double s_d[1];
  __shared__ int s_i[2];
  __shared__ float sh[SMX_NW];
  const int tid = threadIdx.x;
  float v0[SMX_U];
  smx_load(c.P, c.lo, c.hi, v0);
  const unsigned long long h = smx_merged<LVL - 1>(c.ws);
  SmxLevel in;
  if constexpr (LVL == 1) {
    float ps = c.ws->psum[tid & (SMX_G - 1)];
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < SMX_G; ++g) tot += __shfl(ps, g, 64);               // slice order: the same total in every workgroup
    in.cum_above = 0.0; in.target = (double)(a.sp.top_p * tot);             // the descending cumulative sum must EXCEED this
    in.prefix = 0u; in.mask = 0u;
  } else in = c.ws->lv[LVL - 2];
  const LmState s = *c.st;
  if (!smx_drawing(a, s)) return;
  smx_hist_clear(hb);
  const SmxLevel out = smx_walk<LVL - 1>(h, in, shd, s_i, s_d);           // (its barriers publish the cleared histogram)
  if (blockIdx.x == 0 && tid == 0) c.ws->lv[LVL - 1] = out;
  float above = 0.f;
  smx_each(c.P, c.lo, c.hi, v0, [&](int, float p, int) {
    smx_hist_put<LVL>(hb, p, out.prefix, out.mask);
    if (LVL == SMX_LV - 1 && (__float_as_uint(p) & out.mask) > out.prefix) above += p;
  });
  if (LVL == SMX_LV - 1) { above = blk_sum<SMX_NW>(above, sh); if (tid == 0) c.ws->pabove[blockIdx.x] = above; }
  else __syncthreads();
  smx_hist_store<LVL>(c, hb);
}

// 4. the last level's walk, then the inverse-CDF draw over the kept tokens in index order with the caller's uniform: serial over the 32
//    slices, then the whole workgroup resolves the selected slice (thread t owns CH contiguous tokens, a prefix sum finds the first
//    chunk that crosses the goal and its thread walks it in index order)
__global__ __launch_bounds__(SMX_NT) void smx_draw(SmxArgs a) {
  const SmxCtx c = smx_ctx(a);                            // gridDim.x = 1: lo / hi are not used here
  LmState* st = c.st;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int L3 = SMX_LV - 1, NB3 = SMX_NB[L3];
  __shared__ double shd[SMX_NW];
  __shared__ double s_d[1];
  __shared__ float sh[SMX_NW];
  __shared__ int shi[SMX_NW];
  __shared__ float s_f[3];
  __shared__ int s_i[5];
  __shared__ unsigned long long part[SMX_NT];
  __shared__ float pkeep[SMX_G];
  __shared__ int ptie[SMX_G];
  __shared__ float slice[SMX_NT * SMX_U];
  const bool use_top_p = smx_use_top_p(a.sp, a.V);
  // level-3 records: thread t holds bin (t & 127) of the slices 16 (t >> 7) .. + 15 -- for the merged walk AND for the per-slice sums
  const int my_bin = tid & (NB3 - 1), g0 = (tid >> 7) * (SMX_G / 2);
  unsigned long long rec[SMX_G / 2];
#pragma unroll
  for (int g = 0; g < SMX_G / 2; ++g) rec[g] = use_top_p ? c.ws->slab[L3][g0 + g][my_bin] : 0ull;
  const float ps = c.ws->psum[tid & (SMX_G - 1)], pa = c.ws->pabove[tid & (SMX_G - 1)];
  SmxLevel in = c.ws->lv[L3 - 1];
  const LmState s = *st;
  const int pos = s.pos, cur_len = pos + 1;
  if (!smx_drawing(a, s)) {                               // still consuming the prompt (or done): nothing to draw
    if (tid == 0) st->pos = pos + 1;
    return;
  }
  const float u01 = c.uniforms[s.n_gen];
  unsigned thr_bits = 0u;        // keep p > thr, plus `keep_ties` of the p == thr (lowest index first)
  int keep_ties = 0x7fffffff;
  if (use_top_p) {
    unsigned long long hsum = 0ull;
#pragma unroll
    for (int g = 0; g < SMX_G / 2; ++g) hsum += rec[g];
    part[tid] = hsum;
    __syncthreads();
    const int wb = NB3 - 1 - tid;                         // the walk wants bin nb-1-t in thread t
    const unsigned long long h = wb >= 0 ? part[wb] + part[wb + NB3] : 0ull;
    int sel_bin = 0;
    const SmxLevel out = smx_walk<L3>(h, in, shd, s_i, s_d, &sel_bin);
    thr_bits = out.prefix;
    const float thr = __uint_as_float(thr_bits);
    // ties: keep the smallest k >= 1 with cum_above + k*thr > target
    int k = 1;
    if (thr > 0.f) { const double need = (out.target - out.cum_above) / (double)thr; k = need >= 2.0e9 ? 0x7fffffff : (int)floor(need) + 1; if (k < 1) k = 1; }
    keep_ties = k;
    // every slice's kept sum = what lies above the level-3 bin range + its bins above the selected one; its ties = the selected bin's count
    // (thread (g, j) adds bins 16 j .. 16 j + 15 of slice g from LDS)
    __shared__ unsigned long long recs[SMX_G][NB3 + 1];
#pragma unroll
    for (int g = 0; g < SMX_G / 2; ++g) recs[g0 + g][my_bin] = rec[g];
    __syncthreads();
    const int g = tid >> 3, j = tid & 7;
    const float pag = __shfl(pa, g & 31, 64);             // pabove[g] (lane l holds slice l & 31; every lane takes part in the shuffle)
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NB3 / 8; ++q) {
      const int bin = j * (NB3 / 8) + q;
      const unsigned long long r = recs[g][bin];
      if (bin > sel_bin && (r >> SMX_CNT_SHIFT) != 0ull) acc += smx_bin_value(r, in.prefix | (unsigned)bin);
    }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
    if (j == 0) { pkeep[g] = pag + (float)acc; ptie[g] = (int)(recs[g][sel_bin] >> SMX_CNT_SHIFT); }
  } else if (tid < SMX_G) { pkeep[tid] = ps; ptie[tid] = 0; }
  __syncthreads();
  const float thr = __uint_as_float(thr_bits);
  const float* __restrict__ P = c.P;
  if (tid < 64) {
    // the kept total, then the slice holding the goal (ties count lowest index first, up to keep_ties): prefix sums over the 32 slices
    // in lanes 0 .. 31 of the first wave (fixed order: deterministic)
    const bool on = lane < SMX_G;
    const int my_tie = on ? ptie[lane] : 0;
    const float my_keep = on ? pkeep[lane] : 0.f;
    const int tie_incl = wave_incl_scan(my_tie, lane);
    const int nt = __shfl(tie_incl, 63, 64);
    const int kt = use_top_p ? min(keep_ties, nt) : 0;
    const int used = min(kt, tie_incl - my_tie);          // ties already used by the slices before this one
    const int tk = min(my_tie, kt - used);
    const float add = my_keep + (float)tk * thr;
    const float add_incl = wave_incl_scan(add, lane);
    const float goal = u01 * __shfl(add_incl, 63, 64);
    const float cum = add_incl - add;
    const unsigned long long crosses = __ballot(on && cum + add > goal), holds = __ballot(on && add > 0.f);
    // goal >= kept total (rounding): the last slice that holds anything
    const int sel = crosses ? __ffsll((long long)crosses) - 1 : (holds ? 63 - __clzll((long long)holds) : 0);
    if (lane == sel) { s_f[1] = cum; s_f[2] = goal; s_i[1] = sel; s_i[2] = used; s_i[3] = kt; s_i[0] = 0x7fffffff; s_i[4] = -1; }
  }
  __syncthreads();
  const int sel = s_i[1], kt = s_i[3], ties0 = s_i[2];
  const float cum0 = s_f[1], goal = s_f[2];
  const int r_lo = min(a.V, sel * c.sl), r_hi = min(a.V, r_lo + c.sl);
  // the selected slice through LDS (coalesced, all loads in flight), then thread t reads its own contiguous chunk
  const bool staged = c.sl <= SMX_NT * SMX_U;
  if (staged) {
    float v[SMX_U];
    smx_load(P, r_lo, r_hi, v);
#pragma unroll
    for (int u = 0; u < SMX_U; ++u) slice[tid + SMX_NT * u] = v[u];
  }
  __syncthreads();
  auto at = [&](int i) { return staged ? slice[i - r_lo] : P[i]; };
  const int CH = (c.sl + SMX_NT - 1) / SMX_NT;
  const int t_lo = min(r_hi, r_lo + tid * CH), t_hi = min(r_hi, t_lo + CH);
  float tsum = 0.f; int tt = 0; bool any_kept = false;
  for (int i = t_lo; i < t_hi; ++i) {
    const float p = at(i); const unsigned b = __float_as_uint(p);
    if (!use_top_p || b > thr_bits) { tsum += p; any_kept = any_kept || p > 0.f; } else if (b == thr_bits) ++tt;
  }
  const float ex_sum = blk_excl_scan<float, SMX_NW>(tsum, sh);
  const int ex_tie = blk_excl_scan<int, SMX_NW>(tt, shi);
  const int used_before = min(kt, ties0 + ex_tie), used_after = min(kt, ties0 + ex_tie + tt);
  const float before = cum0 + ex_sum + (float)(used_before - ties0) * thr;
  const float after = cum0 + (ex_sum + tsum) + (float)(used_after - ties0) * thr;
  any_kept = any_kept || (used_after > used_before && thr > 0.f);
  if (any_kept) { atomicMax(&s_i[4], tid); if (after > goal) atomicMin(&s_i[0], tid); }
  __syncthreads();
  const int win = s_i[0] != 0x7fffffff ? s_i[0] : s_i[4];      // no chunk crosses (rounding): the last chunk that holds a kept token
  if (tid == (win < 0 ? 0 : win)) {
    int found = -1, last_kept = -1;
    float cum = before; int used = used_before;
    for (int i = t_lo; i < t_hi && found < 0; ++i) {
      const float p = at(i); const unsigned b = __float_as_uint(p);
      float kv = 0.f;
      if (!use_top_p || b > thr_bits) kv = p;
      else if (b == thr_bits && used < kt) { kv = thr; ++used; }
      if (kv > 0.f) { last_kept = i; if (cum + kv > goal) found = i; cum += kv; }
    }
    if (found < 0) found = last_kept >= 0 ? last_kept : min(r_lo, a.V - 1);
    const int next = found;
    const int ng = s.n_gen;
    if (cur_len < a.max_ctx) c.tokens[cur_len] = next;
    st->n_gen = ng + 1;
    bool stop = false;
    for (int k = 0; k < a.sp.n_stop; ++k) stop = stop || next == a.sp.stop_ids[k];
    if (!stop && a.sp.rep_window > 0) {                  // history is updated only for non-stop tokens (OrpheusTTS.swift:304-326)
      int nh = s.n_hist;
      if (nh < a.sp.rep_window) { c.hist[nh] = next; st->n_hist = nh + 1; }
      else { for (int k = 1; k < nh; ++k) c.hist[k - 1] = c.hist[k]; c.hist[nh - 1] = next; }
    }
    if (stop || ng + 1 >= a.sp.max_new_tokens || cur_len + 1 >= a.max_ctx) st->finished = 1;
    st->pos = pos + 1;
  }
  (void)lane;
}

// six launches; ws: one SmxWs per sequence
void lm_sample_launch(hipStream_t s, float* logits, int V, int32_t* tokens, int32_t* hist, const float* uniforms, LmState* st, SmxWs* ws, const mia_lm_sampler& sp,
                      int n_prompt, int max_ctx, int B = 1) {
  const dim3 g(SMX_G, B), one(1, B), blk(SMX_NT);
  const SmxArgs a{logits, V, tokens, hist, uniforms, st, ws, sp, n_prompt, max_ctx};
  hipLaunchKernelGGL(smx_max, g, blk, 0, s, a);
  hipLaunchKernelGGL(smx_exp, g, blk, 0, s, a);
  hipLaunchKernelGGL(smx_level<1>, g, blk, 0, s, a);
  hipLaunchKernelGGL(smx_level<2>, g, blk, 0, s, a);
  hipLaunchKernelGGL(smx_level<3>, g, blk, 0, s, a);
  hipLaunchKernelGGL(smx_draw, one, blk, 0, s, a);
}


// ---- RAS sampler of CosyVoice2 (Qwen2LM.swift:295-321, 433-488): nucleus (top-p 0.8 capped at top-k 25, renormalised, drawn in
// descending-probability order); if the pick already occurs >= win*tau times among the last `win` emitted tokens, redraw from the
// full softmax; while i < min_len an EOS pick is rejected and the whole trial repeated (<= 100 times).  Every categorical draw is
// an inverse CDF with the next caller-provided uniform (u_cursor walks the stream).

constexpr int RAS_NPT = 8;      // logits a thread keeps in registers: V <= 8192 (CosyVoice2: 6561 speech tokens + 3)
__global__ __launch_bounds__(1024) void lm_sample_ras(const float* __restrict__ logits, int V, int32_t* __restrict__ tokens, int32_t* __restrict__ out_tokens,
                                                      const float* __restrict__ uniforms, LmState* __restrict__ st, RasParams rp, int max_ctx) {
  __shared__ float sh[16];
  __shared__ float shv[2][16];
  __shared__ int shi[2][16];
  __shared__ float topv[32];
  __shared__ int topi[32];
  __shared__ int s_tok[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // one workgroup per sequence; the text-length-dependent bounds of the loop live in the sequence's state
  st += blockIdx.x; logits += (int64_t)blockIdx.x * V; tokens += (int64_t)blockIdx.x * max_ctx; out_tokens += (int64_t)blockIdx.x * max_ctx;
  uniforms += (int64_t)blockIdx.x * max_ctx;
  const int pos = st->pos, n_prompt = st->n_embeds;
  const int cur_len = pos + 1;
  if (cur_len < n_prompt || st->finished) { __syncthreads(); if (tid == 0 && !st->finished) st->pos = pos + 1; return; }
  const int step_i = cur_len - n_prompt;                 // loop index i of inferenceLoop
  // the whole row lives in registers (thread t holds ids t, t + 1024, ...): one read of the logits for the statistics and
  // all top-k rounds (re-reading them per round with a `taken` list cost 300 us per token at V = 6564)
  float x[RAS_NPT];
#pragma unroll
  for (int u = 0; u < RAS_NPT; ++u) { const int i = tid + 1024 * u; x[u] = i < V ? logits[i] : -INFINITY; }
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < RAS_NPT; ++u) mx = fmaxf(mx, x[u]);
  mx = blk1024_max(mx, sh);
  float tot = 0.f;
#pragma unroll
  for (int u = 0; u < RAS_NPT; ++u) if (tid + 1024 * u < V) tot += __expf(x[u] - mx);
  tot = blk1024_sum(tot, sh);
  // top-k (k <= 32), ordered by (value desc, index asc).  Fast path: a 3-level radix select (11 | 11 | 10 bits of the order-preserving
  // key, integer LDS atomics, one wave walks the bins) finds the k-th largest key, the <= 32 survivors are gathered and rank-sorted:
  // 8 barriers instead of one per rank (the iterated argmax below measured 2.1 us per round, 52 of the kernel's 60 us).  It is kept as
  // the fallback for the one case the select cannot order by itself: more logits equal to the threshold than slots left for them.
  const int K = rp.top_k < 32 ? rp.top_k : 32;
  __shared__ unsigned rh[3][2048];
  __shared__ int r_sel[4];                               // bin, count above it, count in it, gather cursor
  __shared__ float candv[32];
  __shared__ int candi[32];
  for (int i = tid; i < 3 * 2048; i += 1024) (&rh[0][0])[i] = 0u;
  if (tid == 0) r_sel[3] = 0;
  unsigned key[RAS_NPT];
#pragma unroll
  for (int u = 0; u < RAS_NPT; ++u) {
    const unsigned bits = __float_as_uint(x[u]);
    key[u] = tid + 1024 * u < V ? ((bits & 0x80000000u) ? ~bits : (bits | 0x80000000u)) : 0u;    // 0 sorts below every float
  }
  __syncthreads();
  bool fast = true;
  {
    unsigned prefix = 0u, mask = 0u;
    int need = K, c_thr = 0;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
#pragma unroll
    for (int lvl = 0; lvl < 3; ++lvl) {
      const int shf = shifts[lvl], nb = 1 << widths[lvl];
#pragma unroll
      for (int u = 0; u < RAS_NPT; ++u)
        if (key[u] != 0u && (key[u] & mask) == prefix) atomicAdd(&rh[lvl][(key[u] >> shf) & (unsigned)(nb - 1)], 1u);
      if (tid == 0) r_sel[0] = -1;
      __syncthreads();
      if (wave == 0) {                                   // lane l owns the bins nb-1 - per*l ... nb-per*(l+1), walked downwards
        const int per = nb >> 6;
        int mine = 0;
        for (int j = 0; j < per; ++j) mine += (int)rh[lvl][nb - 1 - (per * lane + j)];
        const int incl = wave_incl_scan(mine, lane), excl = incl - mine;
        if (excl < need && incl >= need) {
          int cum = excl;
          for (int j = 0; j < per; ++j) {
            const int bin = nb - 1 - (per * lane + j), cnt = (int)rh[lvl][bin];
            if (cum + cnt >= need) { r_sel[0] = bin; r_sel[1] = cum; r_sel[2] = cnt; break; }
            cum += cnt;
          }
        }
      }
      __syncthreads();
      if (r_sel[0] < 0) { fast = false; break; }        // fewer than k candidates (cannot happen for k <= V finite logits)
      prefix |= (unsigned)r_sel[0] << shf;
      mask |= (unsigned)(nb - 1) << shf;
      need -= r_sel[1];
      c_thr = r_sel[2];
      __syncthreads();
    }
    if (fast && c_thr > need) fast = false;              // ties at the threshold would have to be split by index: ordered path
    if (fast) {
#pragma unroll
      for (int u = 0; u < RAS_NPT; ++u)
        if (key[u] != 0u && key[u] >= prefix) { const int slot = atomicAdd(&r_sel[3], 1); if (slot < 32) { candv[slot] = x[u]; candi[slot] = tid + 1024 * u; } }
      __syncthreads();
      if (tid < K) {
        const float v = candv[tid]; const int ix = candi[tid];
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += (candv[j] > v || (candv[j] == v && candi[j] < ix)) ? 1 : 0;
        topv[rank] = __expf(v - mx) / tot; topi[rank] = ix;
      }
      __syncthreads();
    }
  }
  if (!fast) {
  for (int r = 0; r < K; ++r) {
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < RAS_NPT; ++u) { const int i = tid + 1024 * u; if (i < V && x[u] > bv) { bv = x[u]; bi = i; } }   // ascending i: first maximum wins
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int buf = r & 1;
    if (lane == 0) { shv[buf][wave] = bv; shi[buf][wave] = bi; }
    __syncthreads();
    float v = shv[buf][0]; int ix = shi[buf][0];
#pragma unroll
    for (int w2 = 1; w2 < 16; ++w2) { const float ov = shv[buf][w2]; const int oi = shi[buf][w2]; if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; } }
    if (tid == 0) { topv[r] = __expf(v - mx) / tot; topi[r] = ix; }
    if (ix != 0x7fffffff && (ix & 1023) == tid) {
#pragma unroll
      for (int u = 0; u < RAS_NPT; ++u) if ((ix >> 10) == u) x[u] = -INFINITY;
    }
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
    const bool ignore_eos = step_i < st->min_len;
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
      if (step_i + 1 >= st->max_len || cur_len + 1 >= max_ctx) st->finished = 1;
      st->pos = pos + 1;
    }
  }
}

// greedy / plain path: advance only (logits are read back by the host)
__global__ void lm_advance(LmState* st) { st[blockIdx.x].pos += 1; }
__global__ void lm_set_pos(LmState* st, int pos) { st->pos = pos; }

// batched prompt pass: act[r][j] = silu(gu[r][2j]) * gu[r][2j+1] from the fp32 GEMM output (same expression as the SK_SWIGLU epilogue)
template <typename T>
__global__ __launch_bounds__(256) void lm_swiglu_rows(const float* __restrict__ gu, uint16_t* __restrict__ act, int64_t n_pairs2) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;      // two outputs per thread
  if (e >= n_pairs2) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(gu + 4 * e);
  const float a = v[0] / (1.0f + __expf(-v[0])) * v[1], b = v[2] / (1.0f + __expf(-v[2])) * v[3];
  *reinterpret_cast<uint32_t*>(act + 2 * e) = pack2<T>(a, b);
}

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

// one launch site for the eight (dtype, head_dim, fused) instances
int lm_launch_attention(mia_lm* m, bool fused, int rows, const void* q, uint16_t* kc, uint16_t* vc, void* att, const int2* rowmap, const float* part, int S, const float* bias) {
  // fused: rows = sequences (row b uses state b and the b-th cache of the layer); otherwise rows = positions of one sequence
  const int64_t seq_stride = (int64_t)m->cfg.n_kv_heads * m->cfg.max_ctx * m->cfg.head_dim;
  const mia_lm_config& c = m->cfg;
  hipStream_t s = m->ctx->stream;
  const int dh = c.head_dim;
  const size_t lds = (size_t)(c.max_ctx + ATT_NW * dh + 2 * ATT_NW + 3 * dh) * 4;
  const float scale = 1.0f / sqrtf((float)dh);
  const dim3 grid(c.n_heads, rows), block(64 * ATT_NW);
#define ATT_GO(TT, DD, FF) hipLaunchKernelGGL((lm_attention<TT, DD, FF>), grid, block, lds, s, (const uint16_t*)q, kc, vc, (uint16_t*)att, m->state, c.n_heads, c.n_kv_heads, c.max_ctx, scale, 0, part, S, bias, m->inv_freq, rows, seq_stride, rowmap)
  const bool f16 = m->dtype == MIA_F16;
  if (dh == 128) { if (fused) { if (f16) ATT_GO(F16, 128, true); else ATT_GO(BF16, 128, true); } else { if (f16) ATT_GO(F16, 128, false); else ATT_GO(BF16, 128, false); } }
  else           { if (fused) { if (f16) ATT_GO(F16, 64, true); else ATT_GO(BF16, 64, true); } else { if (f16) ATT_GO(F16, 64, false); else ATT_GO(BF16, 64, false); } }
#undef ATT_GO
  return 0;
}

int lm_enqueue_step(mia_lm* m, bool sampling, const mia_lm_sampler& sp, int n_prompt, const RasParams* ras = nullptr, int nb = 1) {
  hipStream_t s = m->ctx->stream;
  const mia_lm_config& c = m->cfg;
  const int D = c.hidden, dh = c.head_dim, Nq = c.n_heads * dh, Nk = c.n_kv_heads * dh, Nqkv = Nq + 2 * Nk;
  const bool f16 = m->dtype == MIA_F16;
  // nb sequences = nb rows of every skinny GEMM: the weights are still read once per step
  // RMSNorm rides on the GEMMs (decode.h, SkinnyArgs): the two residual-writing projections (o, down) add into x, emit the next block's
  // activation x * norm weight (16 bit) and per-tile sums of squares; the GEMM that consumes it scales its accumulators by rstd.  Two
  // launches less per layer than "GEMM -> split-K partials -> reduce + norm kernel" (5.1 us each on Orpheus-3B, 56 per token).
  // Up to 4 sequences (the latency-critical case).  Wider batches keep the split-K + reduce / norm chain: without a cross-workgroup split the
  // residual-writing projections run on hidden / 16 workgroups only (32 sentences side by side, Orpheus-3B: 6 400 tokens/s fused, 9 500 split).
  // The choice follows the handle's CAPACITY (mia_lm_set_batch), not the rows of this call: within one capacity every call -- one sequence
  // or many -- runs the same chain, so a sequence's ids do not depend on which batch it sits in.
  const bool fused_norm = (D % 16) == 0 && D <= 8192 && m->B_cap <= 4;
  const int ss_tiles = D / 16;
  float* ss_o = m->ss;                                   // written by o-proj, read by gate|up
  float* ss_d = m->ss + (size_t)ss_tiles * m->B_cap;     // written by down-proj, read by the next q|k|v (or the head)
  struct Norm { const float* ss = nullptr; };            // consumer side: which partial sums (null = activation already normalised)
  auto skinny = [&](const void* A, int64_t lda, const void* W, const void* Wf, const float* bias, void* out, int64_t ldo, int N, int K, int S, int mode,
                    const Q4W* qw = nullptr, const float* ss_in = nullptr, const float* nw = nullptr, float* ss_out = nullptr) {
    SkinnyArgs a{(const uint16_t*)A, lda, (const uint16_t*)W, bias, out, ldo, nullptr, nullptr, nullptr, nb, N, K, S, MIA_ACT_NONE, 0, 0, 0};
    if (ss_in) { a.ss_in = ss_in; a.ss_tiles = ss_tiles; a.ss_dim = D; a.eps = c.rms_eps; }
    if (mode == SK_RESID) { a.xres = m->x; a.nw = nw; a.ss_out = ss_out; }
    if (m->q4 && qw && qw->wfrag) return skinny_gemm_q_launch(a, qw->wfrag, qw->stfrag, m->q_bits, mode, m->dtype, s);
    if (Wf) { a.W = (const uint16_t*)Wf; a.w_frag = 1; }      // same K order and partition as the row-major form: identical results
    return skinny_gemm_launch(a, mode, m->dtype, s);
  };
#define LAUNCH_T(kern, grid, block, lds, ...) do { if (f16) hipLaunchKernelGGL((kern<F16>), grid, block, lds, s, __VA_ARGS__); else hipLaunchKernelGGL((kern<BF16>), grid, block, lds, s, __VA_ARGS__); } while (0)
  LAUNCH_T(lm_embed_norm, dim3(nb), dim3(256), 0, m->tokens, (const uint16_t*)m->embed, (const uint16_t*)m->gen_embed, m->embeds, m->layers[0].in_norm, m->x, (uint16_t*)m->h, m->state, D, c.rms_eps, (const int2*)nullptr, c.max_ctx, c.vocab, m->gen_rows);
  const size_t layer_stride = (size_t)m->B_cap * c.n_kv_heads * c.max_ctx * dh;
  for (int l = 0; l < c.n_layers; ++l) {
    const LmLayer& L = m->layers[l];
    uint16_t* kc = (uint16_t*)m->k_cache + (size_t)l * layer_stride;
    uint16_t* vc = (uint16_t*)m->v_cache + (size_t)l * layer_stride;
    const float* next_norm = l + 1 < c.n_layers ? m->layers[l + 1].in_norm : m->final_norm;
    if (fused_norm) {
      // layer 0 reads the embedding kernel's (normalised) h; later layers the previous down-proj's x * norm weight + its sums of squares
      if (skinny(m->h, D, L.wqkv, L.wqkv_f, nullptr, m->qkv_part, 0, Nqkv, D, m->S_qkv, SK_PARTIAL, &L.q_qkv, l > 0 ? ss_d : nullptr)) return -1;
      lm_launch_attention(m, true, nb, nullptr, kc, vc, m->att, nullptr, m->qkv_part, m->S_qkv, L.bqkv);      // RoPE + cache row + attention
      if (skinny(m->att, Nq, L.wo, L.wo_f, nullptr, m->h, D, D, Nq, 1, SK_RESID, &L.q_o, nullptr, L.post_norm, ss_o)) return -1;
      if (skinny(m->h, D, L.wgu, L.wgu_f, nullptr, m->act, c.inter, 2 * c.inter, D, 1, SK_SWIGLU, &L.q_gu, ss_o)) return -1;
      if (skinny(m->act, c.inter, L.wdown, L.wdown_f, nullptr, m->h, D, D, c.inter, 1, SK_RESID, &L.q_down, nullptr, next_norm, ss_d)) return -1;
      continue;
    }
    if (skinny(m->h, D, L.wqkv, L.wqkv_f, nullptr, m->qkv_part, 0, Nqkv, D, m->S_qkv, SK_PARTIAL, &L.q_qkv)) return -1;
    lm_launch_attention(m, true, nb, nullptr, kc, vc, m->att, nullptr, m->qkv_part, m->S_qkv, L.bqkv);      // RoPE + cache row + attention
    if (skinny(m->att, Nq, L.wo, L.wo_f, nullptr, m->partial, 0, D, Nq, m->S_o, SK_PARTIAL, &L.q_o)) return -1;
    LAUNCH_T(lm_reduce_norm, dim3(nb), dim3(256), 0, m->partial, m->S_o, L.post_norm, m->x, (uint16_t*)m->h, D, c.rms_eps, nb);
    if (skinny(m->h, D, L.wgu, L.wgu_f, nullptr, m->act, c.inter, 2 * c.inter, D, 1, SK_SWIGLU, &L.q_gu)) return -1;
    if (skinny(m->act, c.inter, L.wdown, L.wdown_f, nullptr, m->partial, 0, D, c.inter, m->S_down, SK_PARTIAL, &L.q_down)) return -1;
    LAUNCH_T(lm_reduce_norm, dim3(nb), dim3(256), 0, m->partial, m->S_down, next_norm, m->x, (uint16_t*)m->h, D, c.rms_eps, nb);
  }
#undef LAUNCH_T
  const int HV = m->head_vocab > 0 ? m->head_vocab : c.vocab;
  if (skinny(m->h, D, m->lm_head, m->lm_head_f, m->head_bias, m->logits, HV, HV, D, 1, SK_OUTF32, &m->q_head, fused_norm && c.n_layers > 0 ? ss_d : nullptr)) return -1;
  if (ras) hipLaunchKernelGGL(lm_sample_ras, dim3(nb), dim3(1024), 0, s, m->logits, HV, m->tokens, m->out_tokens, m->uniforms, m->state, *ras, c.max_ctx);
  else if (sampling) lm_sample_launch(s, m->logits, HV, m->tokens, m->hist, m->uniforms, m->state, (SmxWs*)m->smx, sp, n_prompt, c.max_ctx, nb);
  else hipLaunchKernelGGL(lm_advance, dim3(nb), dim3(1), 0, s, m->state);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- batched prompt pass ---------------------------------------------------------------------------
// Positions [pos0, pos0 + P) of the sequence (token ids already in m->tokens, or embedding rows in m->embeds) go through the
// layers as rows of bf16/f16 MFMA GEMMs (gemm.hip) instead of P single-row step graphs: one read of the weights per chunk of
// PF_ROWS positions instead of one per position.  Only the K/V caches are produced -- the caller runs the LAST prompt position
// through the ordinary step (which owns the head and the sampler), so P = n_prompt - 1.  Per-row arithmetic mirrors the step:
// fp32 residual stream, 16-bit RMSNorm output, fp32 q|k|v -> RoPE -> 16-bit, same attention kernel (row r sees keys [0, pos0+r]),
// fp32 gate/up -> silu(g)*u -> 16-bit.  (The reference does the same thing: model(promptIds, cache) is one batched call,
// OrpheusTTS.swift:262-279 / Qwen2LM.swift:353-361.)
constexpr int PF_ROWS = 512;
constexpr int PF_MIN_ROWS = 8;     // below this the step graph is as fast

bool lm_prefill_supported(const mia_lm* m) {
  const bool off = (m->debug_flags & 2) != 0;      // test hook (mia_lm_set_debug)
  const mia_lm_config& c = m->cfg;
  return !off && c.hidden % 64 == 0 && (c.n_heads * c.head_dim) % 64 == 0 && c.inter % 64 == 0;
}

// rows: (sequence, position) pairs, every sequence's positions ascending; after the pass set_pos[b] (if >= 0) becomes state b's position
int lm_prefill_rows(mia_lm* m, const std::vector<int2>& rows, const std::vector<int>& set_pos) {
  mia_ctx* ctx = m->ctx;
  hipStream_t s = ctx->stream;
  const mia_lm_config& c = m->cfg;
  const int D = c.hidden, dh = c.head_dim, Nq = c.n_heads * dh, Nk = c.n_kv_heads * dh, Nqkv = Nq + 2 * Nk, I = c.inter;
  const bool f16 = m->dtype == MIA_F16;
  const size_t b_x = align_up((size_t)PF_ROWS * D * 4, 256), b_h = align_up((size_t)PF_ROWS * D * 2, 256), b_qkv = align_up((size_t)PF_ROWS * Nqkv * 4, 256),
               b_q = align_up((size_t)PF_ROWS * Nq * 2, 256), b_gu = align_up((size_t)PF_ROWS * 2 * I * 4, 256), b_act = align_up((size_t)PF_ROWS * I * 2, 256),
               b_map = align_up((size_t)PF_ROWS * sizeof(int2), 256);
  if (!m->pf_buf) {
    void* p = nullptr;
    if (hipMalloc(&p, b_x + b_h + b_qkv + 2 * b_q + b_gu + b_act + b_map) != hipSuccess) return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "lm: prompt-pass buffers");
    m->allocs.push_back(p); m->pf_buf = (char*)p;
  }
  char* b = m->pf_buf;
  float* x = (float*)b; b += b_x;
  uint16_t* h = (uint16_t*)b; b += b_h;
  float* qkv = (float*)b; b += b_qkv;
  uint16_t* q = (uint16_t*)b; b += b_q;
  uint16_t* att = (uint16_t*)b; b += b_q;
  float* gu = (float*)b; b += b_gu;
  uint16_t* act = (uint16_t*)b; b += b_act;
  int2* rowmap = (int2*)b;
  auto gemm = [&](const void* A, int K, const void* W, const float* bias, float* C, int N, int M, const float* R) {
    GemmArgs g;
    g.A = A; g.lda = K; g.W = W; g.bias = bias; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.out_f32 = 1;
    if (R) { g.R = R; g.ldr = N; }
    if (const char* e = mia_gemm_check(g)) { ctx->err = e; return -1; }
    return mia_gemm_launch(g, m->dtype, s);
  };
#define LAUNCH_T(kern, grid, block, lds, ...) do { if (f16) hipLaunchKernelGGL((kern<F16>), grid, block, lds, s, __VA_ARGS__); else hipLaunchKernelGGL((kern<BF16>), grid, block, lds, s, __VA_ARGS__); } while (0)
  const int P = (int)rows.size();
  const int64_t seq_stride = (int64_t)c.n_kv_heads * c.max_ctx * dh;
  const size_t layer_stride = (size_t)m->B_cap * seq_stride;
  for (int r0 = 0; r0 < P; r0 += PF_ROWS) {
    const int M = std::min(PF_ROWS, P - r0);
    // (pageable source: the copy is staged before the call returns, and the stream orders it behind the previous chunk's kernels)
    MIA_HIP(ctx, hipMemcpyAsync(rowmap, rows.data() + r0, (size_t)M * sizeof(int2), hipMemcpyHostToDevice, s));
    LAUNCH_T(lm_embed_norm, dim3(M), dim3(256), 0, m->tokens, (const uint16_t*)m->embed, (const uint16_t*)m->gen_embed, m->embeds, m->layers[0].in_norm, x, h,
             m->state, D, c.rms_eps, (const int2*)rowmap, c.max_ctx, c.vocab, m->gen_rows);
    for (int l = 0; l < c.n_layers; ++l) {
      const LmLayer& L = m->layers[l];
      uint16_t* kc = (uint16_t*)m->k_cache + (size_t)l * layer_stride;
      uint16_t* vc = (uint16_t*)m->v_cache + (size_t)l * layer_stride;
      if (gemm(h, D, L.wqkv, nullptr, qkv, Nqkv, M, nullptr)) return MIA_ERR_DEVICE;
      const int n_el = (c.n_heads + c.n_kv_heads) * (dh / 2) + Nk;
      LAUNCH_T(lm_rope_cache, dim3((n_el + 255) / 256, M), dim3(256), 0, qkv, 1, L.bqkv, m->inv_freq, q, kc, vc, (const int2*)rowmap, c.n_heads, c.n_kv_heads, dh, c.max_ctx, seq_stride);
      if (l + 1 == c.n_layers) break;            // past its K/V rows the last layer feeds only the head, which the prompt pass skips
      lm_launch_attention(m, false, M, q, kc, vc, att, rowmap, nullptr, 0, nullptr);
      if (gemm(att, Nq, L.wo, nullptr, x, D, M, x)) return MIA_ERR_DEVICE;                // x += att . Wo^T
      LAUNCH_T(lm_reduce_norm, dim3(M), dim3(256), 0, (const float*)nullptr, 0, L.post_norm, x, h, D, c.rms_eps, 1);
      if (gemm(h, D, L.wgu, nullptr, gu, 2 * I, M, nullptr)) return MIA_ERR_DEVICE;
      const int64_t n2 = (int64_t)M * I / 2;
      LAUNCH_T(lm_swiglu_rows, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, gu, act, n2);
      if (gemm(act, I, L.wdown, nullptr, x, D, M, x)) return MIA_ERR_DEVICE;              // x += act . Wdown^T
      LAUNCH_T(lm_reduce_norm, dim3(M), dim3(256), 0, (const float*)nullptr, 0, m->layers[l + 1].in_norm, x, h, D, c.rms_eps, 1);
    }
  }
#undef LAUNCH_T
  for (size_t sq = 0; sq < set_pos.size(); ++sq)
    if (set_pos[sq] >= 0) hipLaunchKernelGGL(lm_set_pos, dim3(1), dim3(1), 0, s, m->state + sq, set_pos[sq]);
  if (hipGetLastError() != hipSuccess) return mia_fail(ctx, MIA_ERR_DEVICE, "lm: prompt-pass launch failed");
  return MIA_OK;
}

// positions [pos0, pos0 + P) of one sequence
int lm_prefill(mia_lm* m, int pos0, int P, int seq = 0) {
  std::vector<int2> rows(P);
  for (int i = 0; i < P; ++i) rows[i] = make_int2(seq, pos0 + i);
  std::vector<int> set_pos(seq + 1, -1);
  set_pos[seq] = pos0 + P;
  return lm_prefill_rows(m, rows, set_pos);
}

int lm_graph(mia_lm* m, int mode, const mia_lm_sampler& sp, const RasParams* ras = nullptr, int nb = 1) {
  mia_ctx* ctx = m->ctx;
  if (m->debug_flags & 1) return 1;                  // test hook (mia_lm_set_debug): launch every step directly
  mia_lm_sampler key{}; RasParams rkey{};
  if (mode == 1) key = sp;
  if (mode == 2) rkey = *ras;
  if (m->graph && m->graph_mode == mode && m->graph_nb == nb && memcmp(&m->graph_sampler, &key, sizeof(key)) == 0 && memcmp(&m->graph_ras, &rkey, sizeof(rkey)) == 0) return 0;
  if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
  hipGraph_t g = nullptr;
  MIA_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  const int erc = lm_enqueue_step(m, mode != 0, sp, -1, mode == 2 ? ras : nullptr, nb);
  hipError_t ce = hipStreamEndCapture(ctx->stream, &g);
  if (erc != 0 || ce != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return mia_fail(ctx, MIA_ERR_DEVICE, "lm: step graph capture failed"); }
  hipError_t ie = hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess) { m->graph = nullptr; return mia_fail(ctx, MIA_ERR_DEVICE, "lm: hipGraphInstantiate failed"); }
  m->graph_sampler = key; m->graph_ras = rkey; m->graph_mode = mode; m->graph_nb = nb;
  return 0;
}

}  // namespace

namespace {

// every per-sequence buffer, for B sequences side by side (rows of the skinny GEMMs; caches [L][B][Hkv][max_ctx][dh])
int lm_alloc_state(mia_lm* m, int B) {
  const mia_lm_config& c = m->cfg;
  const int D = c.hidden, dh = c.head_dim, Nq = c.n_heads * dh, Nk = c.n_kv_heads * dh;
  for (void* p : m->state_allocs) (void)hipFree(p);
  m->state_allocs.clear();
  if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
  bool ok = true;
  auto dev = [&](size_t bytes) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes + 64) != hipSuccess) { ok = false; return nullptr; }
    m->state_allocs.push_back(p);
    return p;
  };
  const size_t kv = (size_t)c.n_layers * B * c.n_kv_heads * c.max_ctx * dh * 2;
  m->k_cache = dev(kv); m->v_cache = dev(kv);
  m->x = (float*)dev((size_t)B * D * 4); m->h = dev((size_t)B * D * 2);
  m->qkv_part = (float*)dev((size_t)4 * B * (Nq + 2 * Nk) * 4); m->q = dev((size_t)B * Nq * 2); m->att = dev((size_t)B * Nq * 2); m->act = dev((size_t)B * c.inter * 2);
  m->ss = (float*)dev((size_t)2 * ((D + 15) / 16) * B * 4);
  m->partial = (float*)dev((size_t)8 * B * D * 4); m->logits = (float*)dev((size_t)B * std::max(c.vocab, m->head_vocab) * 4);
  m->tokens = (int32_t*)dev((size_t)B * c.max_ctx * 4); m->hist = (int32_t*)dev((size_t)B * 64 * 4); m->uniforms = (float*)dev((size_t)B * c.max_ctx * 4);
  m->state = (LmState*)dev(sizeof(LmState) * B); m->smx = dev(sizeof(SmxWs) * B);
  m->embeds = (float*)dev((size_t)B * c.max_ctx * D * 4); m->out_tokens = (int32_t*)dev((size_t)B * c.max_ctx * 4);
  if (!ok) return -1;
  (void)hipMemset(m->k_cache, 0, kv); (void)hipMemset(m->v_cache, 0, kv); (void)hipMemset(m->state, 0, sizeof(LmState) * B);
  m->B_cap = B;
  return 0;
}

}  // namespace

extern "C" void mia_lm_free(mia_lm* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->graph) (void)hipGraphExecDestroy(m->graph);
  for (void* p : m->allocs) (void)hipFree(p);
  for (void* p : m->state_allocs) (void)hipFree(p);
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
  {  // fragment-order copies for the decode step (one device repack per matrix)
    bool ok = true;
    auto frag = [&](const void* src, int N, int K) -> void* {
      if (!ok || !src || K % 32 != 0) return nullptr;
      void* dst = nullptr;
      if (hipMalloc(&dst, (size_t)((N + 15) / 16) * 16 * K * 2) != hipSuccess) { ok = false; return nullptr; }
      m->allocs.push_back(dst);
      if (dec_launch_repack_wfrag(src, dst, N, K, ctx->stream) != 0) ok = false;
      return dst;
    };
    for (LmLayer& ly : m->layers) {
      ly.wqkv_f = frag(ly.wqkv, Nq + 2 * Nk, D); ly.wo_f = frag(ly.wo, D, Nq);
      ly.wgu_f = frag(ly.wgu, 2 * c.inter, D); ly.wdown_f = frag(ly.wdown, D, c.inter);
    }
    m->lm_head_f = frag(m->lm_head, m->head_vocab > 0 ? m->head_vocab : c.vocab, D);
    if (!ok) return fail(m, "fragment-order repack of the step weights failed");
  }
  m->S_qkv = pick_split(D, 4); m->S_o = pick_split(Nq, 4); m->S_down = pick_split(c.inter, 8);
  if (lm_alloc_state(m, 1)) return fail(m, "hipMalloc failed (state buffers)");
  if (hipDeviceSynchronize() != hipSuccess) return fail(m, "device error during upload");
  return m;
}

// ---- MLX-affine 4-bit weights for the decode step (OrpheusWeightLoader.swift:28-60: the reference's default checkpoints are q4, group 64) ----
namespace {

struct Q4Src { const uint32_t* w; const uint16_t* s; const uint16_t* b; };   // one Linear as stored: packed [N][K*bits/32], scales / biases [N][K/64]

float q16_to_f32(uint16_t v, int sdt) {
  if (sdt == MIA_F16) { _Float16 h; memcpy(&h, &v, 2); return (float)h; }
  const uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f;
}

// rows[i] = (tensor index, row): the fused matrix's row i.  Builds the fragment-ordered arrays (layout and arithmetic: skinny_gemm_qi)
// and uploads them.  bits 4 | 8; mag = the 16-bit float the codes are OR-ed into (128 for bf16, 1024 for f16 compute).
bool q_repack(LmLoader& L, const std::vector<Q4Src>& src, const std::vector<std::pair<int, int>>& rows, int K, int bits, int sdt, float mag, Q4W& out) {
  const int N = (int)rows.size(), tiles = (N + 15) / 16, nblk = K / 128, np = bits / 4, cpw = 32 / bits, wpr = K / cpw, gpr = K / 64;
  std::vector<uint32_t> wf((size_t)tiles * nblk * np * 64 * 4);
  std::vector<float> st((size_t)tiles * nblk * 16 * 4);
  const float tmul = mag * (np == 2 ? 17.0f : 1.0f);
  for (int t = 0; t < tiles; ++t)
    for (int r = 0; r < 16; ++r) {
      const int n = std::min(t * 16 + r, N - 1);                      // the last tile repeats its final row (never stored)
      const Q4Src& q = src[rows[n].first];
      const uint32_t* wrow = q.w + (size_t)rows[n].second * wpr;
      const uint16_t* srow = q.s + (size_t)rows[n].second * gpr;
      const uint16_t* brow = q.b + (size_t)rows[n].second * gpr;
      auto code = [&](int k) -> uint32_t { return (wrow[k / cpw] >> ((k % cpw) * bits)) & ((1u << bits) - 1u); };   // MLX packing: little end first
      for (int b = 0; b < nblk; ++b) {
        for (int c = 0; c < 4; ++c)
          for (int stp = 0; stp < 4; ++stp) {
            const int k0 = b * 128 + 32 * stp + 8 * c;
            for (int p = 0; p < np; ++p) {
              uint32_t word = 0;
              for (int i = 0; i < 4; ++i) {
                const uint32_t q0 = (code(k0 + 2 * i) >> (4 * p)) & 15u, q1 = (code(k0 + 2 * i + 1) >> (4 * p)) & 15u;
                word |= (q0 << (4 * i)) | (q1 << (16 + 4 * i));
              }
              wf[((((size_t)t * nblk + b) * np + p) * 64 + 16 * c + r) * 4 + stp] = word;
            }
          }
        for (int g = 0; g < 2; ++g) {
          const float sc = q16_to_f32(srow[2 * b + g], sdt), bi = q16_to_f32(brow[2 * b + g], sdt);
          float* d = &st[(((size_t)t * nblk + b) * 16 + r) * 4 + 2 * g];
          d[0] = sc;
          d[1] = (float)((double)bi - (double)tmul * (double)sc);
        }
      }
    }
  out.wfrag = (uint32_t*)L.dev(wf.size() * 4);
  out.stfrag = (float*)L.dev(st.size() * 4);
  if (!out.wfrag || !out.stfrag) { if (L.err.empty()) L.err = "hipMalloc failed for the packed weights"; return false; }
  if (hipMemcpy(out.wfrag, wf.data(), wf.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(out.stfrag, st.data(), st.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    if (L.err.empty()) L.err = "upload of the packed weights failed";
    return false;
  }
  return true;
}

}  // namespace

// tensors: for every Linear of the step, `<name>.weight` (MIA_U32 packed codes [N][K * bits / 32]), `<name>.scales`, `<name>.biases`
// ([N][K/64], both MIA_F16 or both MIA_BF16), names as in the checkpoint (model.layers.L.self_attn.{q,k,v,o}_proj, mlp.{gate,up,down}_proj,
// model.embed_tokens / lm_head).  The handle must already hold the de-quantised 16-bit weights (mia_lm_load on the expanded
// checkpoint): the batched prompt pass keeps using them, the per-token step switches to the packed form.
extern "C" int mia_lm_attach_quantized(mia_lm* m, const mia_tensor_view* tensors, int n_tensors, int group_size, int bits) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, tensors && n_tensors > 0 && group_size == 64 && (bits == 4 || bits == 8), "lm_attach_quantized: tensors required, group size 64, 4 or 8 bits");
  MIA_CHECK_ARG(ctx, m->q_bits == 0, "lm_attach_quantized: packed weights are already attached to this handle (load a fresh handle to replace them)");
  const mia_lm_config& c = m->cfg;
  const int D = c.hidden, dh = c.head_dim, Nq = c.n_heads * dh, Nk = c.n_kv_heads * dh;
  MIA_CHECK_ARG(ctx, D % 128 == 0 && Nq % 128 == 0 && c.inter % 128 == 0, "lm_attach_quantized: hidden, n_heads*head_dim and inter must be multiples of 128");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  LmLoader L; L.m = m;
  for (int i = 0; i < n_tensors; ++i) if (tensors[i].name && tensors[i].data) L.by_name[tensors[i].name] = &tensors[i];
  int sdt = 0;
  const float mag = m->dtype == MIA_F16 ? 1024.0f : 128.0f;
  auto get = [&](const std::string& p, int N, int K, Q4Src& q) -> bool {
    const mia_tensor_view* w = L.find(p + ".weight"); const mia_tensor_view* s = L.find(p + ".scales"); const mia_tensor_view* b = L.find(p + ".biases");
    if (!w || !s || !b) return false;
    const bool ok = w->dtype == MIA_U32 && w->ndim == 2 && w->shape[0] == N && w->shape[1] == (int64_t)K * bits / 32 && s->ndim == 2 && s->shape[0] == N && s->shape[1] == K / 64 &&
                    b->ndim == 2 && b->shape[0] == N && b->shape[1] == K / 64 && s->dtype == b->dtype && (s->dtype == MIA_F16 || s->dtype == MIA_BF16);
    if (!ok) { if (L.err.empty()) L.err = "'" + p + "' is not a " + std::to_string(bits) + "-bit group-64 Linear of the expected shape (scales / biases must be f16 or bf16)"; return false; }
    if (sdt == 0) sdt = s->dtype;
    if (sdt != s->dtype) { if (L.err.empty()) L.err = "mixed scale dtypes"; return false; }
    q = Q4Src{(const uint32_t*)w->data, (const uint16_t*)s->data, (const uint16_t*)b->data};
    return true;
  };
  auto seq = [](int tensor, int n, std::vector<std::pair<int, int>>& rows) { for (int i = 0; i < n; ++i) rows.push_back({tensor, i}); };
  for (int l = 0; l < c.n_layers && L.err.empty(); ++l) {
    const std::string p = "model.layers." + std::to_string(l);
    LmLayer& ly = m->layers[l];
    std::vector<Q4Src> src(3);
    std::vector<std::pair<int, int>> rows;
    if (get(p + ".self_attn.q_proj", Nq, D, src[0]) && get(p + ".self_attn.k_proj", Nk, D, src[1]) && get(p + ".self_attn.v_proj", Nk, D, src[2])) {
      seq(0, Nq, rows); seq(1, Nk, rows); seq(2, Nk, rows);
      if (!q_repack(L, src, rows, D, bits, sdt, mag, ly.q_qkv)) break;
    }
    src.assign(1, Q4Src{}); rows.clear();
    if (get(p + ".self_attn.o_proj", D, Nq, src[0])) { seq(0, D, rows); if (!q_repack(L, src, rows, Nq, bits, sdt, mag, ly.q_o)) break; }
    src.assign(2, Q4Src{}); rows.clear();
    if (get(p + ".mlp.gate_proj", c.inter, D, src[0]) && get(p + ".mlp.up_proj", c.inter, D, src[1])) {
      for (int i = 0; i < c.inter; ++i) { rows.push_back({0, i}); rows.push_back({1, i}); }      // gate / up rows interleaved like wgu
      if (!q_repack(L, src, rows, D, bits, sdt, mag, ly.q_gu)) break;
    }
    src.assign(1, Q4Src{}); rows.clear();
    if (get(p + ".mlp.down_proj", D, c.inter, src[0])) { seq(0, D, rows); if (!q_repack(L, src, rows, c.inter, bits, sdt, mag, ly.q_down)) break; }
  }
  if (L.err.empty() && m->head_vocab == 0) {     // the tied / plain LM head (the CosyVoice2 speech head stays 16-bit: it is not quantised there)
    std::vector<Q4Src> src(1);
    std::vector<std::pair<int, int>> rows;
    const std::string hp = c.tie_embeddings ? "model.embed_tokens" : "lm_head";
    if (L.find(hp + ".scales", false)) { if (get(hp, c.vocab, D, src[0])) { seq(0, c.vocab, rows); q_repack(L, src, rows, D, bits, sdt, mag, m->q_head); } }
  }
  if (!L.err.empty()) {
    // a partly packed handle must not run: drop every packed pointer (the buffers stay with the handle's allocation list until mia_lm_free)
    for (LmLayer& ly : m->layers) { ly.q_qkv = Q4W{}; ly.q_o = Q4W{}; ly.q_gu = Q4W{}; ly.q_down = Q4W{}; }
    m->q_head = Q4W{};
    const bool oom = L.err.find("hipMalloc") != std::string::npos;
    return mia_fail(ctx, oom ? MIA_ERR_OUT_OF_MEMORY : MIA_ERR_INVALID_ARGUMENT, "lm_attach_quantized: %s", L.err.c_str());
  }
  MIA_HIP(ctx, hipDeviceSynchronize());
  // the packed kernel splits K in 128-input blocks: re-pick the cross-workgroup splits on that granule (the 16-bit step uses the same
  // splits from here on)
  auto split128 = [](int K, int want) { for (int sp = want; sp > 1; --sp) if (K % (128 * sp) == 0) return sp; return 1; };
  m->S_qkv = split128(D, 4); m->S_o = split128(Nq, 4); m->S_down = split128(c.inter, 8);
  m->q_bits = bits;
  m->q4 = true;
  m->graph_mode = -1;          // the captured step holds the 16-bit launches: re-capture
  return MIA_OK;
}

extern "C" int mia_lm_attach_q4(mia_lm* m, const mia_tensor_view* tensors, int n_tensors, int group_size) {
  return mia_lm_attach_quantized(m, tensors, n_tensors, group_size, 4);
}

// switch the step between the packed (1) and the 16-bit (0) weights of a handle that has both (A/B timing, parity tests)
extern "C" int mia_lm_set_debug(mia_lm* m, int flags) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_CHECK_ARG(m->ctx, flags >= 0 && flags <= 3, "lm_set_debug: flags must be 0..3 (got %d)", flags);
  m->debug_flags = flags;
  return MIA_OK;
}

extern "C" int mia_lm_use_q4(mia_lm* m, int on) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_CHECK_ARG(m->ctx, !on || m->q_bits != 0, "lm_use_q4: no packed weights attached");
  m->q4 = on != 0;
  m->graph_mode = -1;
  return MIA_OK;
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
  const int gr = lm_graph(m, 0, none);
  if (gr < 0) return gr;
  int done = 0;
  if (n - 1 >= PF_MIN_ROWS && lm_prefill_supported(m)) {        // all but the last position: K/V only, batched
    if (const int rc = lm_prefill(m, st.pos, n - 1)) return rc;
    done = n - 1;
  }
  for (int i = done; i < n; ++i) {
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
  LmState st{}; st.n_prompt = n_prompt;
  MIA_HIP(ctx, hipMemcpyAsync(m->state, &st, sizeof(st), hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(m->tokens, prompt, (size_t)n_prompt * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(m->uniforms, uniforms, (size_t)sp->max_new_tokens * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  const int gr = lm_graph(m, 1, *sp);
  if (gr < 0) return gr;
  const int total = n_prompt + sp->max_new_tokens - 1;
  int first = 0;
  if (n_prompt - 1 >= PF_MIN_ROWS && lm_prefill_supported(m)) {   // prompt[0 .. n_prompt-2]: K/V only, batched; the last prompt token takes the first step
    if (const int rc = lm_prefill(m, 0, n_prompt - 1)) return rc;
    first = n_prompt - 1;
  }
  for (int step = first; step < total; ++step) {
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

// Sequences side by side: every step reads the weights once for all of them (rows of the same skinny GEMMs), each sequence has
// its own K/V cache, repetition window, uniforms and stop state.  Re-allocates the per-sequence state; 1 restores the default.
extern "C" int mia_lm_set_batch(mia_lm* m, int max_batch) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, max_batch >= 1 && max_batch <= 32, "lm_set_batch: 1 <= max_batch <= 32 (one MFMA row tile)");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (max_batch == m->B_cap) return MIA_OK;
  if (lm_alloc_state(m, max_batch)) return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "lm_set_batch: state buffers for %d sequences", max_batch);
  MIA_HIP(ctx, hipDeviceSynchronize());
  return MIA_OK;
}

// mia_lm_generate for n_seq independent prompts at once (sentence-level batching of OrpheusTTS.generate, OrpheusTTS.swift:179-191: the
// reference runs its sentences one after another).  prompts: the ids of all sequences back to back, prompt_offsets [n_seq + 1];
// uniforms [n_seq][max_new_tokens]; out_tokens [n_seq][max_new_tokens]; n_out [n_seq].  Sequence b's ids equal what mia_lm_generate
// returns for prompt b with uniforms row b (rows of a GEMM do not mix; asserted in tests/test_lm_gpu.py).
extern "C" int mia_lm_generate_batch(mia_lm* m, const int32_t* prompts, const int32_t* prompt_offsets, int n_seq, const mia_lm_sampler* sp,
                                     const float* uniforms, int32_t* out_tokens, int32_t* n_out) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, prompts && prompt_offsets && sp && uniforms && out_tokens && n_out, "lm_generate_batch: null arguments");
  MIA_CHECK_ARG(ctx, n_seq >= 1 && n_seq <= m->B_cap, "lm_generate_batch: n_seq %d exceeds the batch set with mia_lm_set_batch (%d)", n_seq, m->B_cap);
  MIA_CHECK_ARG(ctx, sp->max_new_tokens > 0 && sp->rep_window >= 0 && sp->rep_window <= 64 && sp->n_stop >= 0 && sp->n_stop <= 4, "lm_generate_batch: bad sampler");
  const int C = m->cfg.max_ctx, mn = sp->max_new_tokens;
  int min_prompt = C;
  for (int b = 0; b < n_seq; ++b) {
    const int np_ = prompt_offsets[b + 1] - prompt_offsets[b];
    MIA_CHECK_ARG(ctx, np_ > 0 && np_ + mn <= C, "lm_generate_batch: prompt %d: length %d + max_new_tokens exceeds max_ctx", b, np_);
    for (int i = prompt_offsets[b]; i < prompt_offsets[b + 1]; ++i) MIA_CHECK_ARG(ctx, prompts[i] >= 0 && prompts[i] < m->cfg.vocab, "lm_generate_batch: token %d out of vocabulary", prompts[i]);
    min_prompt = std::min(min_prompt, np_);
  }
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  std::vector<LmState> st(n_seq);
  for (int b = 0; b < n_seq; ++b) {
    const int np_ = prompt_offsets[b + 1] - prompt_offsets[b];
    st[b] = LmState{}; st[b].n_prompt = np_;
    MIA_HIP(ctx, hipMemcpyAsync(m->tokens + (size_t)b * C, prompts + prompt_offsets[b], (size_t)np_ * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(m->uniforms + (size_t)b * C, uniforms + (size_t)b * mn, (size_t)mn * 4, hipMemcpyHostToDevice, s));
  }
  MIA_HIP(ctx, hipMemcpyAsync(m->state, st.data(), sizeof(LmState) * n_seq, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  const int gr = lm_graph(m, 1, *sp, nullptr, n_seq);
  if (gr < 0) return gr;
  // prompts: everything but each prompt's last position through the batched prompt pass, sequence by sequence; when that path is
  // not available every sequence walks its prompt in the step graph (the sampler idles until its own prompt is consumed)
  const bool pre = lm_prefill_supported(m);
  int first_steps = 0;
  if (pre) {                                             // the rows of all prompts share the GEMMs of one prompt pass
    std::vector<int2> rows;
    std::vector<int> set_pos(n_seq, -1);
    for (int b = 0; b < n_seq; ++b) {
      for (int i = 0; i + 1 < st[b].n_prompt; ++i) rows.push_back(make_int2(b, i));
      if (st[b].n_prompt > 1) set_pos[b] = st[b].n_prompt - 1;
    }
    if (!rows.empty()) { if (const int rc = lm_prefill_rows(m, rows, set_pos)) return rc; }
  } else {
    for (int b = 0; b < n_seq; ++b) first_steps = std::max(first_steps, st[b].n_prompt - 1);
  }
  const int total = first_steps + mn;
  for (int step = 0; step < total; ++step) {
    if (gr == 0) MIA_HIP(ctx, hipGraphLaunch(m->graph, s));
    else if (lm_enqueue_step(m, true, *sp, -1, nullptr, n_seq)) return mia_fail(ctx, MIA_ERR_DEVICE, "lm_generate_batch: launch failed");
    if ((step & 15) == 15) {
      MIA_HIP(ctx, hipMemcpyAsync(st.data(), m->state, sizeof(LmState) * n_seq, hipMemcpyDeviceToHost, s));
      MIA_HIP(ctx, hipStreamSynchronize(s));
      bool all = true;
      for (int b = 0; b < n_seq; ++b) all = all && st[b].finished;
      if (all) break;
    }
  }
  MIA_HIP(ctx, hipMemcpyAsync(st.data(), m->state, sizeof(LmState) * n_seq, hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  for (int b = 0; b < n_seq; ++b) {
    n_out[b] = st[b].n_gen;
    if (st[b].n_gen > 0) MIA_HIP(ctx, hipMemcpyAsync(out_tokens + (size_t)b * mn, m->tokens + (size_t)b * C + st[b].n_prompt, (size_t)st[b].n_gen * 4, hipMemcpyDeviceToHost, s));
  }
  MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}

// standalone sampler on caller-provided logits (OrpheusTTS.sampleNextToken, OrpheusTTS.swift:375-470)
extern "C" int mia_sample_top_p(mia_ctx* ctx, const float* logits, int V, const int32_t* history, int n_hist, float rep_penalty, float temperature,
                                float top_p, float uniform, int32_t* out) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, logits && out && V > 0 && n_hist >= 0 && n_hist <= 64, "sample_top_p: bad arguments");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const size_t need = align_up((size_t)V * 4, 256) + 1024 + sizeof(SmxWs);
  char* ws = (char*)mia_workspace(ctx, need);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_logits = (float*)ws;
  char* tail = ws + align_up((size_t)V * 4, 256);
  int32_t* d_tok = (int32_t*)tail;            // [2]: tokens[0..1]
  int32_t* d_hist = (int32_t*)(tail + 64);    // [64]
  float* d_u = (float*)(tail + 64 + 256);
  LmState* d_st = (LmState*)(tail + 64 + 256 + 64);
  SmxWs* d_smx = (SmxWs*)(tail + 1024);
  LmState st{0, n_hist, 0, 0};
  hipStream_t s = ctx->stream;
  MIA_HIP(ctx, hipMemcpyAsync(d_logits, logits, (size_t)V * 4, hipMemcpyHostToDevice, s));
  if (n_hist) MIA_HIP(ctx, hipMemcpyAsync(d_hist, history, (size_t)n_hist * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_u, &uniform, 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(d_st, &st, sizeof(st), hipMemcpyHostToDevice, s));
  mia_lm_sampler sp{}; sp.temperature = temperature; sp.top_p = top_p; sp.rep_penalty = rep_penalty; sp.rep_window = 0; sp.max_new_tokens = 1;
  lm_sample_launch(s, d_logits, V, d_tok, d_hist, d_u, d_st, d_smx, sp, 1, 2);
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
  MIA_CHECK_ARG(ctx, m->head_vocab <= 1024 * RAS_NPT && rp->top_k <= m->head_vocab, "lm_generate_ras: llm_decoder has %d rows; the RAS sampler holds at most %d", m->head_vocab, 1024 * RAS_NPT);
  MIA_CHECK_ARG(ctx, rp->top_k > 0 && rp->top_k <= 32 && rp->win >= 0 && rp->win <= 64 && rp->eos >= 0 && rp->eos < m->head_vocab, "lm_generate_ras: bad sampler parameters");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  LmState st{}; st.n_embeds = n_prompt; st.min_len = rp->min_len; st.max_len = rp->max_len;
  MIA_HIP(ctx, hipMemcpyAsync(m->state, &st, sizeof(st), hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipMemcpyAsync(m->embeds, prompt_embeds, (size_t)n_prompt * m->cfg.hidden * 4, hipMemcpyHostToDevice, s));
  const int nu = std::min(n_uniforms, m->cfg.max_ctx);
  MIA_HIP(ctx, hipMemcpyAsync(m->uniforms, uniforms, (size_t)nu * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  RasParams r{rp->top_p, rp->top_k, rp->win, rp->tau, rp->eos, 0, 0, nu};      // (the length bounds travel in the state: one graph serves every text length)
  mia_lm_sampler none{};
  const int gr = lm_graph(m, 2, none, &r);
  if (gr < 0) return gr;
  const int total = n_prompt + rp->max_len - 1;
  int first = 0;
  if (n_prompt - 1 >= PF_MIN_ROWS && lm_prefill_supported(m)) {   // [sos, text, task, prompt speech) rows except the last: K/V only, batched
    if (const int rc = lm_prefill(m, 0, n_prompt - 1)) return rc;
    first = n_prompt - 1;
  }
  for (int step = first; step < total; ++step) {
    if (gr == 0) MIA_HIP(ctx, hipGraphLaunch(m->graph, s));
    else if (lm_enqueue_step(m, true, none, n_prompt, &r)) return mia_fail(ctx, MIA_ERR_DEVICE, "lm_generate_ras: launch failed");
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

// mia_lm_generate_ras for n_seq utterances side by side (see mia_lm_generate_batch).  prompt_embeds: the rows of all prompts back to back
// [prompt_offsets[n_seq]][hidden]; rp [n_seq]: top_p / top_k / win / tau / eos must agree, min_len / max_len are per utterance;
// uniforms [n_seq][n_uniforms]; out_tokens [n_seq][out_stride] with out_stride >= max(max_len) + 1; n_out [n_seq].
extern "C" int mia_lm_generate_ras_batch(mia_lm* m, const float* prompt_embeds, const int32_t* prompt_offsets, int n_seq, const mia_ras_params* rp,
                                         const float* uniforms, int n_uniforms, int32_t* out_tokens, int out_stride, int32_t* n_out) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, prompt_embeds && prompt_offsets && rp && uniforms && n_uniforms > 0 && out_tokens && n_out, "lm_generate_ras_batch: null arguments");
  MIA_CHECK_ARG(ctx, n_seq >= 1 && n_seq <= m->B_cap, "lm_generate_ras_batch: n_seq %d exceeds the batch set with mia_lm_set_batch (%d)", n_seq, m->B_cap);
  MIA_CHECK_ARG(ctx, m->gen_embed && m->head_vocab > 0 && m->head_vocab <= 1024 * RAS_NPT, "lm_generate_ras_batch: model has no (or too large a) speech head");
  const int C = m->cfg.max_ctx, D = m->cfg.hidden;
  int max_len = 0;
  for (int b = 0; b < n_seq; ++b) {
    const int np_ = prompt_offsets[b + 1] - prompt_offsets[b];
    MIA_CHECK_ARG(ctx, np_ > 0 && rp[b].max_len > 0 && np_ + rp[b].max_len <= C, "lm_generate_ras_batch: utterance %d: prompt + max_len exceeds max_ctx", b);
    MIA_CHECK_ARG(ctx, rp[b].top_p == rp[0].top_p && rp[b].top_k == rp[0].top_k && rp[b].win == rp[0].win && rp[b].tau == rp[0].tau && rp[b].eos == rp[0].eos,
                  "lm_generate_ras_batch: top_p / top_k / win / tau / eos must be the same for all utterances");
    max_len = std::max(max_len, rp[b].max_len);
  }
  MIA_CHECK_ARG(ctx, rp[0].top_k > 0 && rp[0].top_k <= 32 && rp[0].top_k <= m->head_vocab && rp[0].win >= 0 && rp[0].win <= 64 && rp[0].eos >= 0 && rp[0].eos < m->head_vocab,
                "lm_generate_ras_batch: bad sampler parameters");
  MIA_CHECK_ARG(ctx, out_stride >= max_len + 1, "lm_generate_ras_batch: out_stride must hold max_len + 1 ids");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int nu = std::min(n_uniforms, C);
  std::vector<LmState> st(n_seq);
  for (int b = 0; b < n_seq; ++b) {
    const int np_ = prompt_offsets[b + 1] - prompt_offsets[b];
    st[b] = LmState{}; st[b].n_embeds = np_; st[b].min_len = rp[b].min_len; st[b].max_len = rp[b].max_len;
    MIA_HIP(ctx, hipMemcpyAsync(m->embeds + (size_t)b * C * D, prompt_embeds + (size_t)prompt_offsets[b] * D, (size_t)np_ * D * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(m->uniforms + (size_t)b * C, uniforms + (size_t)b * n_uniforms, (size_t)nu * 4, hipMemcpyHostToDevice, s));
  }
  MIA_HIP(ctx, hipMemcpyAsync(m->state, st.data(), sizeof(LmState) * n_seq, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  RasParams r{rp[0].top_p, rp[0].top_k, rp[0].win, rp[0].tau, rp[0].eos, 0, 0, nu};
  mia_lm_sampler none{};
  const int gr = lm_graph(m, 2, none, &r, n_seq);
  if (gr < 0) return gr;
  int first_steps = 0;
  if (lm_prefill_supported(m)) {
    std::vector<int2> rows;
    std::vector<int> set_pos(n_seq, -1);
    for (int b = 0; b < n_seq; ++b) {
      for (int i = 0; i + 1 < st[b].n_embeds; ++i) rows.push_back(make_int2(b, i));
      if (st[b].n_embeds > 1) set_pos[b] = st[b].n_embeds - 1;
    }
    if (!rows.empty()) { if (const int rc = lm_prefill_rows(m, rows, set_pos)) return rc; }
  } else {
    for (int b = 0; b < n_seq; ++b) first_steps = std::max(first_steps, st[b].n_embeds - 1);
  }
  const int total = first_steps + max_len;
  for (int step = 0; step < total; ++step) {
    if (gr == 0) MIA_HIP(ctx, hipGraphLaunch(m->graph, s));
    else if (lm_enqueue_step(m, true, none, -1, &r, n_seq)) return mia_fail(ctx, MIA_ERR_DEVICE, "lm_generate_ras_batch: launch failed");
    if ((step & 15) == 15) {
      MIA_HIP(ctx, hipMemcpyAsync(st.data(), m->state, sizeof(LmState) * n_seq, hipMemcpyDeviceToHost, s));
      MIA_HIP(ctx, hipStreamSynchronize(s));
      bool all = true;
      for (int b = 0; b < n_seq; ++b) all = all && st[b].finished;
      if (all) break;
    }
  }
  MIA_HIP(ctx, hipMemcpyAsync(st.data(), m->state, sizeof(LmState) * n_seq, hipMemcpyDeviceToHost, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));
  for (int b = 0; b < n_seq; ++b) {
    n_out[b] = st[b].n_out;
    if (st[b].n_out > 0) MIA_HIP(ctx, hipMemcpyAsync(out_tokens + (size_t)b * out_stride, m->out_tokens + (size_t)b * C, (size_t)st[b].n_out * 4, hipMemcpyDeviceToHost, s));
  }
  MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}
