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
#include <cstdlib>

#include <type_traits>
#include "decode.h"

// ------------------------------------------------------------------------------------------------
// Row kernels of the decode step: one 256-thread workgroup per clip, the row (D <= 2048 floats) lives in
// registers as float4, statistics via wave shuffles + one LDS hop.
// ------------------------------------------------------------------------------------------------
constexpr int ROW_NV = 2;   // float4 per thread: D <= 256 * 4 * ROW_NV

__device__ __forceinline__ float block256_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// h is written in activation FRAGMENT order (decode.h): the row is the A operand of the step's next skinny GEMM
template <typename T>
__device__ __forceinline__ void row_layernorm_store(const f32x4 (&v)[ROW_NV], int nv, int D, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, uint16_t* __restrict__ h, int row, float* sh) {
  const int tid = threadIdx.x;
  // gamma / beta do not depend on the statistics: fetch them first, so their L2 round trip overlaps the two block reductions
  f32x4 gm[ROW_NV], bt[ROW_NV];
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const bool in = tid + 256 * i < nv;
    gm[i] = in ? *reinterpret_cast<const f32x4*>(gamma + 4 * (tid + 256 * i)) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bt[i] = in ? *reinterpret_cast<const f32x4*>(beta + 4 * (tid + 256 * i)) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) if (tid + 256 * i < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = block256_sum(s, sh) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i)
    if (tid + 256 * i < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(block256_sum(q, sh) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = tid + 256 * i;
    if (c >= nv) continue;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * gm[i][j] + bt[i][j];
    *reinterpret_cast<u32x2*>(h + afrag_index(row, 4 * c, D)) = (u32x2){pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
  }
}

// x[b] = E[token[b][pos]] + P[pos];  h[b] = LN(x[b])          (TextDecoder.swift:67 + first attn_ln)
template <typename T>
__global__ __launch_bounds__(256) void dec_embed_ln(const int32_t* __restrict__ tokens, const uint16_t* __restrict__ emb,
                                                    const float* __restrict__ pos_emb, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ x,
                                                    uint16_t* __restrict__ h, const int32_t* __restrict__ pos_arr, int D, int n_ctx) {
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int pos = pos_arr[b];
  const int tok = tokens[b * n_ctx + pos];
  const uint16_t* e = emb + (int64_t)tok * D;
  const float* p = pos_emb + (int64_t)pos * D;
  float* xr = x + (int64_t)b * D;
  const int nv = D >> 2;
  f32x4 v[ROW_NV];
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      const s16x4 ev = *reinterpret_cast<const s16x4*>(e + 4 * c);
      const f32x4 pv = *reinterpret_cast<const f32x4*>(p + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = T::to_f32((uint16_t)ev[j]) + pv[j];
      *reinterpret_cast<f32x4*>(xr + 4 * c) = v[i];
    } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  row_layernorm_store<T>(v, nv, D, gamma, beta, h, b, sh);
}

// x[b] += bias + sum_s partial[s][b];  h[b] = LN(x[b])        (residual add + next LayerNorm, fixed-order split-K sum)
// (A one-wave-per-row form -- no LDS hop, no barrier, the statistics as plain DPP wave reductions -- was measured: 1.6 us SLOWER per
// launch; a single wave issuing the row's 30-40 loads takes longer than four waves with two barriers.)
template <typename T>
__global__ __launch_bounds__(256) void dec_reduce_ln(const float* __restrict__ partial, int S, int B, const float* __restrict__ bias,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ x, uint16_t* __restrict__ h, int D) {
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float* xr = x + (int64_t)b * D;
  const int nv = D >> 2;
  f32x4 v[ROW_NV];
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      f32x4 a = *reinterpret_cast<const f32x4*>(xr + 4 * c);
      const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += bs[j];
      // split-K slices come from other XCDs (L2 misses): issue every load before the first add.  S <= 4; a slot k >= S
      // re-reads slice S-1 and is dropped, so the loads are unconditional and the sum keeps its fixed order.
      f32x4 pv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pv[k] = *reinterpret_cast<const f32x4*>(partial + ((int64_t)(k < S ? k : S - 1) * B + b) * D + 4 * c);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < S) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += pv[k][j];
        }
      v[i] = a;
      *reinterpret_cast<f32x4*>(xr + 4 * c) = a;
    } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  row_layernorm_store<T>(v, nv, D, gamma, beta, h, b, sh);
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM for decode (M <= 32 rows per block-z): out[m][n] = sum_k A[m][k] W[n][k].
// One wave per 16 output columns and K-slice; weights go HBM -> VGPR (each weight byte is read exactly once),
// activations come from L2; v_mfma_f32_16x16x32 with W as the row operand so a lane owns 4 consecutive n.
// ------------------------------------------------------------------------------------------------
// NT = 16-column tiles per wave (activation fragments are reused NT times: NT=4 for the 51866-wide logits GEMM,
// where the L2->CU activation traffic would otherwise be twice the HBM weight traffic); KB = K-steps per batch.
// NW = waves per workgroup, each taking 1/NW of the K range (intra-block split-K, summed through LDS in wave order: deterministic).
// With one wave per CU the weight stream is latency-bound (12 KB in flight per CU); NW = 4 quadruples the loads in flight.
// (Cross-workgroup tickets were tried for fusing the split-K reduction + LayerNorm into this kernel: same-address device-scope atomics
// from ~640 workgroups on 8 XCDs cost ~60 us per launch -- far more than the 5 us kernel boundary they would remove.)
// fixed-order cross-wave sum of the NW waves' K-slices: waves 1.. park their fragments in LDS, wave 0 adds them in wave order and
// is the only one to return true (it owns the epilogue)
template <int NT, int NW>
__device__ __forceinline__ bool skinny_wave_reduce(f32x4 (&acc)[NT][2], int wave, int lane) {
  if (NW == 1) return true;
  __shared__ f32x4 red[(NW > 1 ? NW - 1 : 1) * NT * 2 * 64];
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) red[(((wave - 1) * NT + t) * 2 + mt) * 64 + lane] = acc[t][mt];
  }
  __syncthreads();
  if (wave > 0) return false;
#pragma unroll
  for (int w2 = 1; w2 < NW; ++w2)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f32x4 o = red[(((w2 - 1) * NT + t) * 2 + mt) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][mt][j] += o[j];
      }
  return true;
}

// rstd of the (up to 32) rows of this workgroup from the producer's per-tile partial sums of squares (SkinnyArgs::ss_in): computed by
// wave 0 at the START of the kernel -- its loads ride under the weight stream -- into LDS; the epilogue (wave 0 again, behind the
// reduction barrier when the workgroup has one) reads rs[row].  Lanes take tiles t = lane, lane + 64, ...; fixed-order sums.
__device__ __forceinline__ void skinny_rstd_prepare(const SkinnyArgs& a, float* rs, int m0, int wave, int lane) {
  if (!a.ss_in || wave != 0) return;
  // (the LM step uses this form with at most 4 rows -- lm.hip keeps the unfused chain for wider batches -- so a row at a time is fine:
  // all of a row's loads are issued together, one wave reduction per row)
  const int rows = a.M - m0 < 32 ? a.M - m0 : 32;
  for (int j = 0; j < rows; ++j) {
    float p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int t = lane + 64 * u; p[u] = t < a.ss_tiles ? a.ss_in[(int64_t)t * a.M + m0 + j] : 0.f; }   // tiles <= 512
    float v = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    v = wave_sum(v);
    if (lane == 0) rs[j] = rsqrtf(v / (float)a.ss_dim + a.eps);
  }
}

// lane holds C[m = m0 + mt*16 + r][n = n0 + 16t + 4c + j]
template <typename T, int MODE, int NT>
__device__ __forceinline__ void skinny_epilogue(const SkinnyArgs& a, const f32x4 (&acc)[NT][2], int n0, int m0, int split, int lane) {
  const int r = lane & 15, c = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + 4 * c;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + mt * 16 + r;
      if (m >= a.M) continue;
      const f32x4 av = acc[t][mt];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (n + j >= a.N) continue;
        float v = av[j];
        if (MODE == SK_PARTIAL) {
          reinterpret_cast<float*>(a.out)[((int64_t)split * a.M + m) * a.N + n + j] = v;
          continue;
        }
        if (a.bias) v += a.bias[n + j];
        if (MODE == SK_SWIGLU) {   // interleaved rows: even column = gate, odd column = up
          if (j & 1) continue;
          float u = av[j + 1];
          if (a.bias) u += a.bias[n + j + 1];
          const float sg = v / (1.0f + __expf(-v));
          reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + ((n + j) >> 1)] = T::from_f32(sg * u);
          continue;
        }
        if (a.act == MIA_ACT_GELU) v = gelu_erf(v);
        if (MODE == SK_OUTF32) reinterpret_cast<float*>(a.out)[(int64_t)m * a.ldo + n + j] = v;
        else if (MODE == SK_OUT16) reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + n + j] = T::from_f32(v);
        else {  // SK_QKV: [0,D) -> q, [D,2D) -> self K cache, [2D,3D) -> self V cache at position pos
          const int nn = n + j;
          if (nn < a.D) reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + nn] = T::from_f32(v);
          else {
            const int hd = (nn - a.D) % a.D, h = hd >> 6, d = hd & 63;
            uint16_t* cache = nn < 2 * a.D ? a.cache_k : a.cache_v;
            cache[(((int64_t)m * a.H + h) * a.n_ctx + a.pos[m]) * 64 + d] = T::from_f32(v);
          }
        }
      }
    }
  }
}

// The epilogue's small dependent loads (the lane's 4 bias values per tile, the cache position of its two rows) are issued by
// skinny_prefetch at the START of the kernel, next to the operand loads: fetched in the epilogue they added an L2 round trip to every
// biased GEMM of the chain, after the reduction barrier where nothing hides it.
template <int NT>
struct SkinnyPre { float bs[NT][4]; int pos[2]; float c1[NT][4], c2[NT][4]; };

template <int MODE, int NT>
__device__ __forceinline__ SkinnyPre<NT> skinny_prefetch(const SkinnyArgs& a, int n0, int m0, int lane) {
  SkinnyPre<NT> p;
  const int r = lane & 15, c = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + 4 * c;
#pragma unroll
    for (int j = 0; j < 4; ++j) p.bs[t][j] = (MODE != SK_PARTIAL && a.bias && n + j < a.N) ? a.bias[n + j] : 0.f;
    // LayerNorm fold constants of a consumer (SkinnyArgs::c1 / c2): one 16-byte load each (n is a multiple of 4, the arrays hipMalloc'ed)
    if (a.c1 && n + 3 < a.N) {
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(a.c1 + n), v2 = *reinterpret_cast<const f32x4*>(a.c2 + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) { p.c1[t][j] = v1[j]; p.c2[t][j] = v2[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = a.c1 && n + j < a.N;
        p.c1[t][j] = in ? a.c1[n + j] : 0.f;
        p.c2[t][j] = in ? a.c2[n + j] : 0.f;
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + mt * 16 + r;
    p.pos[mt] = (MODE == SK_QKV && m < a.M) ? a.pos[m] : 0;
  }
  return p;
}

template <typename T, int MODE, int NT>
__device__ __forceinline__ void skinny_epilogue_v(const SkinnyArgs& a, const f32x4 (&acc)[NT][2], const SkinnyPre<NT>& pre, int n0, int m0, int split,
                                                  int lane, const float* st = nullptr) {
  static_assert(MODE != SK_SWIGLU, "SK_SWIGLU is stored by skinny_store");
  const int r = lane & 15, c = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + 4 * c;
    if (n >= a.N) continue;
    const bool full = n + 3 < a.N;
    const float (&bs)[4] = pre.bs[t];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + mt * 16 + r;
      if (m >= a.M) continue;
      f32x4 v = acc[t][mt];
      if (a.c1 && st) {      // the activation was x * gamma (SK_RESID producer): LN(x) W^T = rstd (acc - mean c1) + c2
        const float mean = st[2 * (mt * 16 + r)], rstd = st[2 * (mt * 16 + r) + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rstd * (v[j] - mean * pre.c1[t][j]) + pre.c2[t][j];
      }
      if (MODE == SK_PARTIAL) {
        float* dst = reinterpret_cast<float*>(a.out) + ((int64_t)split * a.M + m) * a.N + n;
        if (full && (a.N & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
        else { for (int j = 0; j < 4; ++j) if (n + j < a.N) dst[j] = v[j]; }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += bs[j]; if (a.act == MIA_ACT_GELU) v[j] = gelu_erf(v[j]); }
      if (MODE == SK_OUTF32) {
        float* dst = reinterpret_cast<float*>(a.out) + (int64_t)m * a.ldo + n;
        if (full && (a.ldo & 1) == 0) {
          *reinterpret_cast<f32x2*>(dst) = (f32x2){v[0], v[1]};
          *reinterpret_cast<f32x2*>(dst + 2) = (f32x2){v[2], v[3]};
        } else { for (int j = 0; j < 4; ++j) if (n + j < a.N) dst[j] = v[j]; }
        continue;
      }
      const u32x2 pk = (u32x2){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
      uint16_t* o16 = reinterpret_cast<uint16_t*>(a.out);
      if (MODE == SK_OUT16) {
        if (a.out_frag) { *reinterpret_cast<u32x2*>(o16 + afrag_index(m, n, a.N)) = pk; continue; }   // host-checked: N % 16 == 0
        uint16_t* dst = o16 + (int64_t)m * a.ldo + n;
        if (full && (a.ldo & 3) == 0) *reinterpret_cast<u32x2*>(dst) = pk;
        else { for (int j = 0; j < 4; ++j) if (n + j < a.N) dst[j] = T::from_f32(v[j]); }
        continue;
      }
      // SK_QKV (host-checked: D % 64 == 0, N == 3 D): [0,D) -> q row-major, [D,2D) -> self K cache, [2D,3D) -> self V cache at pos[m]
      if (n < a.D) { *reinterpret_cast<u32x2*>(o16 + (int64_t)m * a.ldo + n) = pk; continue; }
      const int hd = (n - a.D) % a.D, h = hd >> 6, d = hd & 63;
      uint16_t* cache = n < 2 * a.D ? a.cache_k : a.cache_v;
      *reinterpret_cast<u32x2*>(cache + (((int64_t)m * a.H + h) * a.n_ctx + pre.pos[mt]) * 64 + d) = pk;
    }
  }
}

// Row-major-activation kernels (the LM step): the same vector stores.  SK_SWIGLU: a lane's 4 consecutive columns are (gate, up, gate, up)
// -> two outputs, one 4-byte store; SK_QKV keeps the scalar form (the Whisper step, its only user, runs the fragment-order kernels).
template <typename T, int MODE, int NT>
__device__ __forceinline__ void skinny_store(const SkinnyArgs& a, f32x4 (&acc)[NT][2], int n0, int m0, int split, int lane, const float* rs = nullptr) {
  const int r = lane & 15, c = lane >> 4;
  if (a.ss_in && rs) {       // the activation was stored un-normalised (SK_RESID producer): scale every row by its rstd first
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int mr = mt * 16 + r;
      const float sc = m0 + mr < a.M ? rs[mr] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][mt][j] *= sc;
    }
  }
  if constexpr (MODE == SK_RESID) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + 4 * c;                 // host-checked: N % 16 == 0
      const f32x4 wv = *reinterpret_cast<const f32x4*>(a.nw + n);
      f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + mt * 16 + r;
        float q = 0.f;
        if (m < a.M) {
          float* xp = a.xres + (int64_t)m * a.N + n;
          f32x4 x = *reinterpret_cast<const f32x4*>(xp);
#pragma unroll
          for (int j = 0; j < 4; ++j) { x[j] += acc[t][mt][j] + bv[j]; q += x[j] * x[j]; }
          *reinterpret_cast<f32x4*>(xp) = x;
          *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + (int64_t)m * a.ldo + n) = (u32x2){pack2<T>(x[0] * wv[0], x[1] * wv[1]), pack2<T>(x[2] * wv[2], x[3] * wv[3])};
        }
        // the tile's 16 columns live in the 4 lanes r, r + 16, r + 32, r + 48: fixed-order sum (c = 0, 1, 2, 3)
        const float q1 = __shfl(q, r + 16, 64), q2 = __shfl(q, r + 32, 64), q3 = __shfl(q, r + 48, 64);
        if (c == 0 && m < a.M) a.ss_out[(int64_t)((n0 >> 4) + t) * a.M + m] = ((q + q1) + q2) + q3;
      }
    }
  } else if constexpr (MODE == SK_QKV) {
    skinny_epilogue<T, MODE, NT>(a, acc, n0, m0, split, lane);
  } else if constexpr (MODE == SK_SWIGLU) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = n0 + 16 * t + 4 * c;
      if (n >= a.N) continue;
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < a.N) bs[j] = a.bias[n + j];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + mt * 16 + r;
        if (m >= a.M) continue;
        float o[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float v = acc[t][mt][2 * q] + bs[2 * q], u = acc[t][mt][2 * q + 1] + bs[2 * q + 1];
          o[q] = (v / (1.0f + __expf(-v))) * u;
        }
        uint16_t* dst = reinterpret_cast<uint16_t*>(a.out) + (int64_t)m * a.ldo + (n >> 1);
        if (n + 3 < a.N && (a.ldo & 1) == 0) *reinterpret_cast<uint32_t*>(dst) = pack2<T>(o[0], o[1]);
        else { dst[0] = T::from_f32(o[0]); if (n + 2 < a.N) dst[1] = T::from_f32(o[1]); }
      }
    }
  } else {
    skinny_epilogue_v<T, MODE, NT>(a, acc, skinny_prefetch<MODE, NT>(a, n0, m0, lane), n0, m0, split, lane);
  }
}

// The fragment-order weight matrix of a Whisper-step GEMM as a buffer resource: 16-byte loads at 32-bit byte offsets whose cache policy
// is the instruction's immediate `aux` operand (0 = default, 2 = non-temporal).
struct WFragBuf {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t rs;
  const uint16_t* base;
  __device__ __forceinline__ explicit WFragBuf(const SkinnyArgs& a) : base(a.W) {
    const unsigned bytes = (unsigned)(((a.N + 15) >> 4) << 4) * (unsigned)a.K * 2u;          // wave-uniform
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), (short)0, (int)bytes, 0x00020000);
  }
  // the activation fragments of the same GEMM ([ceil(M / 32)][K / 32][2][64 lanes][8])
  __device__ __forceinline__ WFragBuf(const SkinnyArgs& a, int) : base(a.A) {
    const unsigned bytes = (unsigned)(((a.M + 31) >> 5) << 5) * (unsigned)a.K * 2u;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.A), (short)0, (int)bytes, 0x00020000);
  }
  // the LM step's weight matrix: row-major [N][K] or fragment order (rows padded to 16)
  __device__ __forceinline__ WFragBuf(const SkinnyArgs& a, bool frag) : base(a.W) {
    const unsigned rows = frag ? (unsigned)(((a.N + 15) >> 4) << 4) : (unsigned)a.N;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), (short)0, (int)(rows * (unsigned)a.K * 2u), 0x00020000);
  }
  // the LM step's row-major activations [M][lda]
  struct RowMajorA {};
  __device__ __forceinline__ WFragBuf(const SkinnyArgs& a, RowMajorA) : base(a.A) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.A), (short)0, (int)((unsigned)a.M * (unsigned)a.lda * 2u), 0x00020000);
  }
  __device__ __forceinline__ uint32_t offset(const uint16_t* p) const { return (uint32_t)((const char*)p - (const char*)base); }
  template <int AUX>
  __device__ __forceinline__ s16x8 load(uint32_t byte_off) const {
    return __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, AUX));
  }
};

template <typename T, int MODE, int NT, int KB, int NW>
__global__ __launch_bounds__(64 * NW) void dec_skinny_gemm(SkinnyArgs a) {
  __shared__ float rs[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * (16 * NT);
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * 32;
  const int Kc = a.K / (a.S * NW);
  const int kbeg = (split * NW + wave) * Kc;
  const int r = lane & 15, c = lane >> 4;
  // weights: row-major [N][K] (a K-step of a lane = 16 bytes of row r at column 8 c: 16 rows x 64 B per wave instruction), or in
  // fragment order (decode.h: one contiguous 1 KB per wave instruction); ws = elements between two K-steps of a lane
  const int ws = a.w_frag ? 512 : 32;
  const uint16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (a.w_frag) {
      const int tiles = (a.N + 15) >> 4;
      int tile = (n0 >> 4) + t; tile = tile < tiles ? tile : tiles - 1;
      wp[t] = a.W + (((int64_t)tile * (a.K >> 5) + (kbeg >> 5)) * 64 + lane) * 8;
    } else {
      int wn = n0 + 16 * t + r; wn = wn < a.N ? wn : a.N - 1;
      wp[t] = a.W + (int64_t)wn * a.K + kbeg + 8 * c;
    }
  }
  // weight loads as buffer loads at 32-bit byte offsets (no 64-bit address arithmetic per load; see dec_skinny_fflat)
  const WFragBuf wbuf(a, a.w_frag != 0);
  uint32_t wo[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wo[t] = wbuf.offset(wp[t]);
  const uint32_t wsb = (uint32_t)ws * 2u;
  int am0 = m0 + r; am0 = am0 < a.M ? am0 : a.M - 1;
  int am1 = m0 + 16 + r; am1 = am1 < a.M ? am1 : a.M - 1;
  const uint16_t* ap0 = a.A + (int64_t)am0 * a.lda + kbeg + 8 * c;
  const uint16_t* ap1 = a.A + (int64_t)am1 * a.lda + kbeg + 8 * c;
  // Only the lanes whose row exists fetch activations (the others feed zeros: their accumulator columns are never stored).  At one
  // sequence that is 4 of 64 lanes in the first MFMA half and none in the second: a 16-byte wave load costs the address unit per
  // ACTIVE lane, and with every lane fetching (rows clamped to the last one) the activation fragments were two of the three load
  // instructions of every K-step.
  const bool av0 = m0 + r < a.M, av1 = m0 + 16 + r < a.M;
  const s16x8 zfrag = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
  const WFragBuf abuf(a, WFragBuf::RowMajorA{});
  const uint32_t ab0 = abuf.offset(ap0), ab1 = abuf.offset(ap1);
  auto lda0 = [&](int k) -> s16x8 { s16x8 v = zfrag; if (av0) v = abuf.template load<0>(ab0 + 2u * (uint32_t)k); return v; };
  auto lda1 = [&](int k) -> s16x8 { s16x8 v = zfrag; if (av1) v = abuf.template load<0>(ab1 + 2u * (uint32_t)k); return v; };
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  // Software-pipelined register double buffer: batch i+1 (KB K-steps) is issued before the MFMAs of batch i, so two
  // batches of independent 16-byte loads stay in flight per lane (weights are HBM-once traffic).
  struct Batch { s16x8 w[KB][NT], a0[KB], a1[KB]; };
  auto load_batch = [&](Batch& t, int k) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
#pragma unroll
      for (int n = 0; n < NT; ++n) t.w[u][n] = wbuf.template load<0>(wo[n] + (uint32_t)((k >> 5) + u) * wsb);
      t.a0[u] = lda0(k + 32 * u);
      t.a1[u] = lda1(k + 32 * u);
    }
  };
  auto mma_batch = [&](const Batch& t) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[n][0] = T::mfma16(t.w[u][n], t.a0[u], acc[n][0]);
        acc[n][1] = T::mfma16(t.w[u][n], t.a1[u], acc[n][1]);
      }
  };
  constexpr int KSTEP = 32 * KB;
  const int nb = Kc / KSTEP;
  // The K-steps left over after the whole batches (< KB of them).  On a SHORT slice (at most 3 batches: the ring's fourth register
  // batch is idle) their loads are issued up front into that batch -- fetched after the main loop, one step at a time, they were up
  // to KB - 1 extra dependent memory round trips at the end of every launch (Qwen2-0.5B: K / S = 224 = one batch of 4 steps + 3 such
  // steps, in each of its 72 projection launches per token: 0.876 -> 0.80 ms per token).  Long slices keep the trailing loop: a fifth
  // register batch would cost the 64-thread form its occupancy (measured on Orpheus-3B: 1.96 -> 2.15 ms per token).
  const int ktail = nb * KSTEP, rem = (Kc - ktail) >> 5;
  const bool early_tail = nb <= 3 && rem > 0;
  Batch b0, b1, b2, b3;
  if (early_tail) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
      if (u < rem) {
#pragma unroll
        for (int n = 0; n < NT; ++n) b3.w[u][n] = wbuf.template load<0>(wo[n] + (uint32_t)((ktail >> 5) + u) * wsb);
        b3.a0[u] = lda0(ktail + 32 * u);
        b3.a1[u] = lda1(ktail + 32 * u);
      }
  }
  if (nb > 0) {
    // ring of four register batches (static names: runtime-indexed vector arrays would go to scratch): three batches of
    // loads are always in flight behind the batch being multiplied
    load_batch(b0, 0);
    if (nb > 1) load_batch(b1, KSTEP);
    if (nb > 2) load_batch(b2, 2 * KSTEP);
    skinny_rstd_prepare(a, rs, m0, wave, lane);      // behind the first three batches of loads: its own loads ride under the weight stream
    for (int i = 0; i < nb; i += 4) {
      if (i + 3 < nb) load_batch(b3, (i + 3) * KSTEP);
      mma_batch(b0);
      if (i + 1 >= nb) break;
      if (i + 4 < nb) load_batch(b0, (i + 4) * KSTEP);
      mma_batch(b1);
      if (i + 2 >= nb) break;
      if (i + 5 < nb) load_batch(b1, (i + 5) * KSTEP);
      mma_batch(b2);
      if (i + 3 >= nb) break;
      if (i + 6 < nb) load_batch(b2, (i + 6) * KSTEP);
      mma_batch(b3);
    }
  } else skinny_rstd_prepare(a, rs, m0, wave, lane);
  if (early_tail) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
      if (u < rem) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[n][0] = T::mfma16(b3.w[u][n], b3.a0[u], acc[n][0]);
          acc[n][1] = T::mfma16(b3.w[u][n], b3.a1[u], acc[n][1]);
        }
      }
  } else {
    for (int k = ktail; k < Kc; k += 32) {
      const s16x8 fa0 = lda0(k);
      const s16x8 fa1 = lda1(k);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const s16x8 fw = wbuf.template load<0>(wo[n] + (uint32_t)(k >> 5) * wsb);
        acc[n][0] = T::mfma16(fw, fa0, acc[n][0]);
        acc[n][1] = T::mfma16(fw, fa1, acc[n][1]);
      }
    }
  }
  if (!skinny_wave_reduce<NT, NW>(acc, wave, lane)) return;
  skinny_store<T, MODE, NT>(a, acc, n0, m0, split, lane, rs);
}

// The same GEMM for a SHORT K-slice per wave (exactly NSTEP K-steps of 32, host-checked: K == S * NW * 32 * NSTEP): every operand
// load of the wave is issued before the first MFMA, so the kernel pays ONE memory round trip (the ring above pays one per refill:
// two to three on a 160- or 320-wide slice, ~1.5 us each on a step that is a chain of such kernels).  Weights are read once per
// step and never again before 1.3 GB of other traffic has passed: non-temporal loads keep them from displacing the activations in
// L2 / MALL.  Same K order per wave and same wave-order reduction as dec_skinny_gemm: results are bit-identical to it.
template <typename T, int MODE, int NT, int NSTEP, int NW>
__global__ __launch_bounds__(64 * NW) void dec_skinny_flat(SkinnyArgs a) {
  __shared__ float rs[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * (16 * NT);
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * 32;
  constexpr int Kc = 32 * NSTEP;
  const int kbeg = (split * NW + wave) * Kc;
  const int r = lane & 15, c = lane >> 4;
  const int ws = a.w_frag ? 512 : 32;     // weights row-major or in fragment order: see dec_skinny_gemm
  const uint16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (a.w_frag) {
      const int tiles = (a.N + 15) >> 4;
      int tile = (n0 >> 4) + t; tile = tile < tiles ? tile : tiles - 1;
      wp[t] = a.W + (((int64_t)tile * (a.K >> 5) + (kbeg >> 5)) * 64 + lane) * 8;
    } else {
      int wn = n0 + 16 * t + r; wn = wn < a.N ? wn : a.N - 1;
      wp[t] = a.W + (int64_t)wn * a.K + kbeg + 8 * c;
    }
  }
  const WFragBuf wbuf(a, a.w_frag != 0);
  uint32_t wo[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wo[t] = wbuf.offset(wp[t]);
  const uint32_t wsb = (uint32_t)ws * 2u;
  int am0 = m0 + r; am0 = am0 < a.M ? am0 : a.M - 1;
  int am1 = m0 + 16 + r; am1 = am1 < a.M ? am1 : a.M - 1;
  const uint16_t* ap0 = a.A + (int64_t)am0 * a.lda + kbeg + 8 * c;
  const uint16_t* ap1 = a.A + (int64_t)am1 * a.lda + kbeg + 8 * c;
  const bool av0 = m0 + r < a.M, av1 = m0 + 16 + r < a.M;     // only lanes whose row exists fetch activations (see dec_skinny_gemm)
  const s16x8 zfrag = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
  s16x8 fw[NSTEP][NT], fa0[NSTEP], fa1[NSTEP];
#pragma unroll
  for (int u = 0; u < NSTEP; ++u) {
#pragma unroll
    for (int n = 0; n < NT; ++n) fw[u][n] = wbuf.template load<2>(wo[n] + (uint32_t)u * wsb);
    fa0[u] = zfrag; fa1[u] = zfrag;
    if (av0) fa0[u] = *reinterpret_cast<const s16x8*>(ap0 + 32 * u);
    if (av1) fa1[u] = *reinterpret_cast<const s16x8*>(ap1 + 32 * u);
  }
  __builtin_amdgcn_sched_barrier(0);   // keep every load ahead of the first MFMA (the scheduler would otherwise trade them for registers)
  skinny_rstd_prepare(a, rs, m0, wave, lane);
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int u = 0; u < NSTEP; ++u)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      acc[n][0] = T::mfma16(fw[u][n], fa0[u], acc[n][0]);
      acc[n][1] = T::mfma16(fw[u][n], fa1[u], acc[n][1]);
    }
  if (!skinny_wave_reduce<NT, NW>(acc, wave, lane)) return;
  skinny_store<T, MODE, NT>(a, acc, n0, m0, split, lane, rs);
}

// c1[n] = sum_k W[n][k] gamma[k], c2[n] = sum_k W[n][k] beta[k]: one wave per row of a row-major 16-bit matrix (load time, once per Linear)
template <typename T>
__global__ __launch_bounds__(256) void dec_lnfold(const uint16_t* __restrict__ w, int N, int K, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ c1, float* __restrict__ c2,
                                                  const float* __restrict__ bias) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const uint16_t* wr = w + (int64_t)row * K;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < K; k += 64) { const float v = T::to_f32(wr[k]); s1 = fmaf(v, gamma[k], s1); s2 = fmaf(v, beta[k], s2); }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { c1[row] = s1; c2[row] = s2 + (bias ? bias[row] : 0.f); }       // bias: the encoder's form, c2 + the Linear's own bias
}

int dec_launch_lnfold(const void* w16, int N, int K, const float* gamma, const float* beta, float* c1, float* c2, int dtype, hipStream_t s, const float* bias) {
  if (!w16 || !gamma || !beta || !c1 || !c2 || N <= 0 || K <= 0) return -1;
  if (dtype == MIA_F16) hipLaunchKernelGGL(dec_lnfold<F16>, dim3((N + 3) / 4), dim3(256), 0, s, (const uint16_t*)w16, N, K, gamma, beta, c1, c2, bias);
  else hipLaunchKernelGGL(dec_lnfold<BF16>, dim3((N + 3) / 4), dim3(256), 0, s, (const uint16_t*)w16, N, K, gamma, beta, c1, c2, bias);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// The Whisper step's skinny GEMMs: both operands in MFMA-fragment order (decode.h), so every wave load instruction is one
// contiguous 1 KB, and an epilogue that stores a lane's 4 consecutive columns as one 8- or 16-byte word.
// Measured on the fc1 shape (tools/micro/skinny_probe.hip, N 5120, K 1280, 32 rows, HBM-cold weights): row-major operands + scalar
// stores 11.9 us; weights in fragment order 10.0; activations too 8.3; without the scalar epilogue 5.8; 13 MB streamed by a kernel
// that does nothing else 4.1.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void repack_wfrag(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int N, int K) {
  // one thread per 16-byte fragment chunk: (tile, kstep, lane) <- row 16 tile + (lane & 15), columns 32 kstep + 8 (lane >> 4) .. +7
  const int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int ksteps = K >> 5;
  const int64_t total = (int64_t)((N + 15) >> 4) * ksteps * 64;
  if (chunk >= total) return;
  const int lane = (int)(chunk & 63);
  const int64_t tk = chunk >> 6;
  const int kstep = (int)(tk % ksteps), tile = (int)(tk / ksteps);
  const int n = tile * 16 + (lane & 15), k = kstep * 32 + 8 * (lane >> 4);
  s16x8 v = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) v = *reinterpret_cast<const s16x8*>(src + (int64_t)n * K + k);
  *reinterpret_cast<s16x8*>(dst + chunk * 8) = v;
}

int dec_launch_repack_wfrag(const void* src, void* dst, int N, int K, hipStream_t s) {
  if (N <= 0 || K <= 0 || K % 32 != 0) return -1;
  const int64_t total = (int64_t)((N + 15) / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(repack_wfrag<BF16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, N, K);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// lane holds C[m = m0 + mt*16 + r][n = n0 + 16t + 4c + j], j = 0..3: one vector store per (t, mt)
// short K slice per wave (NSTEP K-steps, host-checked K == 32 * S * NW * NSTEP): every load ahead of the first MFMA.  Weights are
// loaded non-temporal (measured against default-policy loads, hoping the 184 MB of layer weights would stay in the 256 MB MALL from
// step to step: they do not, 0.424 vs 0.418 ms per step)
// LayerNorm carried across the step's GEMM chain (decode.h): (mean, rstd) of the workgroup's rows from the producer's per-tile
// (sum x, sum x^2) pairs.  The NW waves share the rows (wave w takes rows w, w + NW, ...), lanes take tiles t = lane, lane + 64, ...
// (fixed-order sums); results go to LDS st[row][2] and are read by the epilogue behind the reduction barrier (NW == 1: same wave).
template <int NW>
struct LnStat {
  // 64 lanes = 4 row groups x 16 tile lanes; (wave, row group) = one of NW * 4 slots, each owning RPS consecutive rows.  A lane sums
  // its tiles t = l16, l16 + 16, ... locally in that order, the 16 tile lanes of a group are then added by one DPP row reduction:
  // every load of the wave is in flight at once and the summation order is fixed.  issue() runs FIRST in the kernel -- its loads are the
  // oldest of the wave, so finish() can wait for them alone (vmcnt counts in order) while the weight stream issued after them is still in
  // flight; a row-at-a-time loop paid one memory round trip per row (3.5 us on the step's 5 us GEMMs), and loads issued behind the
  // weights made the wave drain its whole queue before the first MFMA.
  static constexpr int RPS = NW >= 8 ? 1 : 8 / NW;             // rows per slot: 32 rows over min(NW * 4, 32) slots
  static constexpr int MAXU = 8;                               // tiles <= 128 (D <= 2048)
  f32x2 v[MAXU][RPS];
  int j0, rows;
  bool on;
  __device__ __forceinline__ void issue(const SkinnyArgs& a, int m0, int wave, int lane) {
    on = a.ss_in != nullptr && a.c1 != nullptr;
    const int rg = lane >> 4, l16 = lane & 15;
    j0 = (wave * 4 + rg) * RPS;
    rows = a.M - m0 < 32 ? a.M - m0 : 32;
#pragma unroll
    for (int u = 0; u < MAXU; ++u)
#pragma unroll
      for (int i = 0; i < RPS; ++i) v[u][i] = (f32x2){0.f, 0.f};
    if (!on) return;
    const f32x2* ss = reinterpret_cast<const f32x2*>(a.ss_in);
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int t = l16 + 16 * u;
      if (t < a.ss_tiles) {
        if (RPS >= 2 && (a.M & 1) == 0) {                      // two rows per 16-byte load (row pairs are 16-byte aligned when M is even)
#pragma unroll
          for (int i = 0; i + 1 < RPS; i += 2)
            if (j0 + i < rows) {
              const f32x4 p = *reinterpret_cast<const f32x4*>(ss + (int64_t)t * a.M + m0 + j0 + i);
              v[u][i] = (f32x2){p[0], p[1]}; v[u][i + 1] = (f32x2){p[2], p[3]};     // (row j + 1 >= rows: never stored)
            }
        } else {
#pragma unroll
          for (int i = 0; i < RPS; ++i)
            if (j0 + i < rows) v[u][i] = ss[(int64_t)t * a.M + m0 + j0 + i];
        }
      }
    }
  }
  __device__ __forceinline__ void finish(const SkinnyArgs& a, float* st, int lane) {
    if (!on) return;
    const int l16 = lane & 15;
#pragma unroll
    for (int i = 0; i < RPS; ++i) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int u = 0; u < MAXU; ++u) { v1 += v[u][i][0]; v2 += v[u][i][1]; }
      v1 += dpp_f32<0xB1>(v1); v1 += dpp_f32<0x4E>(v1); v1 += dpp_f32<0x141>(v1); v1 += dpp_f32<0x140>(v1);
      v2 += dpp_f32<0xB1>(v2); v2 += dpp_f32<0x4E>(v2); v2 += dpp_f32<0x141>(v2); v2 += dpp_f32<0x140>(v2);
      const int j = j0 + i;
      if (l16 == 0 && j < rows) {
        const float mean = v1 / (float)a.ss_dim;
        const float var = fmaxf(v2 / (float)a.ss_dim - mean * mean, 0.f);
        st[2 * j] = mean; st[2 * j + 1] = rsqrtf(var + a.eps);
      }
    }
  }
};

// SK_RESID epilogue of the fragment-order kernels: x[m][n] += acc + bias in place (fp32 residual stream, row stride N); the next
// block's activation x * gamma (its LayerNorm gain; mean / rstd / beta are applied by the consumer) in activation FRAGMENT order;
// per-tile (sum x, sum x^2) pairs.  Lane holds C[m = m0 + 16 mt + r][n = n0 + 16 t + 4 c + j].
template <typename T, int NT>
__device__ __forceinline__ void skinny_resid_frag(const SkinnyArgs& a, const f32x4 (&acc)[NT][2], int n0, int m0, int lane) {
  const int r = lane & 15, c = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + 4 * c;               // host-checked: N % 32 == 0
    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.nw + n);
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = m0 + mt * 16 + r;
      float s1 = 0.f, s2 = 0.f;
      if (m < a.M) {
        float* xp = a.xres + (int64_t)m * a.N + n;
        f32x4 x = *reinterpret_cast<const f32x4*>(xp);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] += acc[t][mt][j] + bv[j]; s1 += x[j]; s2 += x[j] * x[j]; }
        *reinterpret_cast<f32x4*>(xp) = x;
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.out) + afrag_index(m, n, a.N)) =
            (u32x2){pack2<T>(x[0] * gv[0], x[1] * gv[1]), pack2<T>(x[2] * gv[2], x[3] * gv[3])};
      }
      // the tile's 16 columns live in lanes r, r + 16, r + 32, r + 48: fixed-order sums (c = 0, 1, 2, 3)
      const float a1 = __shfl(s1, r + 16, 64), a2 = __shfl(s1, r + 32, 64), a3 = __shfl(s1, r + 48, 64);
      const float b1 = __shfl(s2, r + 16, 64), b2 = __shfl(s2, r + 32, 64), b3 = __shfl(s2, r + 48, 64);
      if (c == 0 && m < a.M)
        *reinterpret_cast<f32x2*>(a.ss_out + ((int64_t)((n0 >> 4) + t) * a.M + m) * 2) = (f32x2){((s1 + a1) + a2) + a3, ((s2 + b1) + b2) + b3};
    }
  }
}

template <typename T, int MODE, int NT, int NSTEP, int NW>
__global__ __launch_bounds__(64 * NW) void dec_skinny_fflat(SkinnyArgs a) {
  __shared__ float st[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int split = blockIdx.y, z = blockIdx.z;
  LnStat<NW> lnstat;
  lnstat.issue(a, z * 32, wave, lane);
  const int ksteps = a.K >> 5, tiles = (a.N + 15) >> 4;
  const int ks0 = (split * NW + wave) * NSTEP;
  const uint16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int tile = blockIdx.x * NT + t; tile = tile < tiles ? tile : tiles - 1;
    wp[t] = a.W + (((int64_t)tile * ksteps + ks0) * 64 + lane) * 8;
  }
  const uint16_t* ap = a.A + ((((int64_t)z * ksteps + ks0) * 2) * 64 + lane) * 8;
  const SkinnyPre<NT> pre = skinny_prefetch<MODE, NT>(a, blockIdx.x * (16 * NT), z * 32, lane);
  s16x8 fw[NSTEP][NT], fa0[NSTEP], fa1[NSTEP];
  // Weights are read once per step: non-temporal, unless other decode loops stream the same copy at the same time (a.w_keep: then the
  // second and third reader mostly hit the Infinity Cache; measured with 3 replicas: + 2.2 % audio-s/s, and - 3.6 % for a lone loop,
  // whose 316 MB per step cycle through the 256 MB cache without a hit).  The cache policy is an instruction bit: two copies of the loop.
  // (`?:` or if / else between a plain and a non-temporal load of ONE pointer is hoisted by LLVM into a single plain load -- both
  //  forms then ran cacheable.  The policy is therefore the immediate `aux` operand of a buffer load, which cannot be merged: 0 = default,
  //  2 = nt.)
  const WFragBuf wb(a);
  const WFragBuf ab(a, 0);
  const uint32_t ao = ab.offset(ap);
  uint32_t wo[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wo[t] = wb.offset(wp[t]);
  auto load_all = [&](auto keep_tag) {
    constexpr int AUX = decltype(keep_tag)::value ? 0 : 2;
#pragma unroll
    for (int u = 0; u < NSTEP; ++u) {
#pragma unroll
      for (int n = 0; n < NT; ++n) fw[u][n] = wb.template load<AUX>(wo[n] + 1024u * u);
      fa0[u] = ab.template load<0>(ao + 2048u * u);
      fa1[u] = ab.template load<0>(ao + 2048u * u + 1024u);
    }
  };
  if (a.w_keep) load_all(std::true_type{}); else load_all(std::false_type{});
  __builtin_amdgcn_sched_barrier(0);   // keep every load ahead of the first MFMA
  lnstat.finish(a, st, lane);          // waits for the statistics alone; the operand loads issued after them stay in flight
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int u = 0; u < NSTEP; ++u)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      acc[n][0] = T::mfma16(fw[u][n], fa0[u], acc[n][0]);
      acc[n][1] = T::mfma16(fw[u][n], fa1[u], acc[n][1]);
    }
  if (!skinny_wave_reduce<NT, NW>(acc, wave, lane)) return;
  if constexpr (MODE == SK_RESID) skinny_resid_frag<T, NT>(a, acc, blockIdx.x * (16 * NT), z * 32, lane);
  else skinny_epilogue_v<T, MODE, NT>(a, acc, pre, blockIdx.x * (16 * NT), z * 32, split, lane, st);
}

// any K slice per wave (K == 32 * S * NW * steps): a ring of four register batches of KB K-steps, three in flight
template <typename T, int MODE, int NT, int KB, int NW>
__global__ __launch_bounds__(64 * NW) void dec_skinny_fring(SkinnyArgs a) {
  __shared__ float st[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int split = blockIdx.y, z = blockIdx.z;
  LnStat<NW> lnstat;
  lnstat.issue(a, z * 32, wave, lane);
  const int ksteps = a.K >> 5, tiles = (a.N + 15) >> 4;
  const int per_wave = ksteps / (a.S * NW);
  const int ks0 = (split * NW + wave) * per_wave;
  const uint16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int tile = blockIdx.x * NT + t; tile = tile < tiles ? tile : tiles - 1;
    wp[t] = a.W + (((int64_t)tile * ksteps + ks0) * 64 + lane) * 8;
  }
  const uint16_t* ap = a.A + ((((int64_t)z * ksteps + ks0) * 2) * 64 + lane) * 8;
  const SkinnyPre<NT> pre = skinny_prefetch<MODE, NT>(a, blockIdx.x * (16 * NT), z * 32, lane);
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  struct Batch { s16x8 w[KB][NT], a0[KB], a1[KB]; };
  const bool keep = a.w_keep != 0;      // cache policy of the weight loads (see dec_skinny_fflat); wave-uniform, one test per batch
  const WFragBuf wb(a);
  const WFragBuf ab(a, 0);
  const uint32_t ao = ab.offset(ap);
  uint32_t wo[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wo[t] = wb.offset(wp[t]);
  auto load_batch = [&](Batch& t, int ks) {
    auto go = [&](auto keep_tag) {
      constexpr int AUX = decltype(keep_tag)::value ? 0 : 2;
#pragma unroll
      for (int u = 0; u < KB; ++u) {
#pragma unroll
        for (int n = 0; n < NT; ++n) t.w[u][n] = wb.template load<AUX>(wo[n] + 1024u * (uint32_t)(ks + u));
        t.a0[u] = ab.template load<0>(ao + 2048u * (uint32_t)(ks + u));
        t.a1[u] = ab.template load<0>(ao + 2048u * (uint32_t)(ks + u) + 1024u);
      }
    };
    if (keep) go(std::true_type{}); else go(std::false_type{});
  };
  auto mma_batch = [&](const Batch& t) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[n][0] = T::mfma16(t.w[u][n], t.a0[u], acc[n][0]);
        acc[n][1] = T::mfma16(t.w[u][n], t.a1[u], acc[n][1]);
      }
  };
  const int nb = per_wave / KB;
  int ks = 0;
  if (nb > 0) {
    Batch b0, b1, b2, b3;
    load_batch(b0, 0);
    if (nb > 1) load_batch(b1, KB);
    if (nb > 2) load_batch(b2, 2 * KB);
    lnstat.finish(a, st, lane);          // the statistics were issued first: this waits for them alone
    for (int i = 0; i < nb; i += 4) {
      if (i + 3 < nb) load_batch(b3, (i + 3) * KB);
      mma_batch(b0);
      if (i + 1 >= nb) break;
      if (i + 4 < nb) load_batch(b0, (i + 4) * KB);
      mma_batch(b1);
      if (i + 2 >= nb) break;
      if (i + 5 < nb) load_batch(b1, (i + 5) * KB);
      mma_batch(b2);
      if (i + 3 >= nb) break;
      if (i + 6 < nb) load_batch(b2, (i + 6) * KB);
      mma_batch(b3);
    }
    ks = nb * KB;
  } else lnstat.finish(a, st, lane);
  for (; ks < per_wave; ++ks) {
    const s16x8 fa0 = *reinterpret_cast<const s16x8*>(ap + 1024 * ks);
    const s16x8 fa1 = *reinterpret_cast<const s16x8*>(ap + 1024 * ks + 512);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const s16x8 fw = *reinterpret_cast<const s16x8*>(wp[n] + 512 * ks);
      acc[n][0] = T::mfma16(fw, fa0, acc[n][0]);
      acc[n][1] = T::mfma16(fw, fa1, acc[n][1]);
    }
  }
  if (!skinny_wave_reduce<NT, NW>(acc, wave, lane)) return;
  if constexpr (MODE == SK_RESID) skinny_resid_frag<T, NT>(a, acc, blockIdx.x * (16 * NT), z * 32, lane);
  else skinny_epilogue_v<T, MODE, NT>(a, acc, pre, blockIdx.x * (16 * NT), z * 32, split, lane, st);
}

// ------------------------------------------------------------------------------------------------
// The same skinny GEMM on MLX-affine quantised weights (group 64: w = scale * code + bias; 4- or 8-bit codes), multiplied PACKED: the
// step streams ~5 (or ~9) bits per weight from HBM instead of 16.  Replaces MLX's quantizedMatmul on the reference's default
// checkpoints (TTS/Orpheus/TTSEngine/OrpheusWeightLoader.swift:28-60, STT/Whisper/WhisperModel.swift:189-200; 4- and 8-bit:
// Models/TranscriptionResult.swift:162-198).
//
// Arithmetic (the one MLX's own qmv / qmm kernels use: scale * sum(code * x) + bias * sum(x) per group): the MFMA runs on the integer
// CODES, which are exact in 16-bit floating point, and scale / bias are applied once per 64-input group to the group's partial sums:
//     y[m][n] = sum_g ( s[n][g] * sum_{k in g} code[n][k] a[m][k]  +  b[n][g] * sum_{k in g} a[m][k] )
// -- no per-weight de-quantisation at all.  A 4-bit code becomes a 16-bit float by OR-ing it into the mantissa of a magic constant
// (bf16 0x4300 | q = 128 + q, f16 0x6400 | q = 1024 + q): with the nibbles stored so that (word >> 4 i) & 0x000f000f isolates the
// codes of K-values 2 i and 2 i + 1, a lane's 8 MFMA operand values cost 7 VALU instructions (the first form of this kernel expanded
// fmaf(scale, code, bias) and re-rounded per weight: 28 to 32 instructions per MFMA, and at one wave per SIMD the kernel ran at the VALU
// issue latency -- 14 us for the 27 MB gate|up matrix of Orpheus-3B, 1.9 TB/s).  The magic offset is removed in the group fix-up:
// sum (MAG + q) a = MAG A + sum q a, so y += s P + t A with t = b - MAG s (fp32, built at attach time) and A = sum_k a[m][k] -- itself
// an MFMA with an all-ones operand, shared by the tiles of a wave.  An 8-bit code is two 4-bit planes, q = 16 hi + lo: the hi plane goes
// through the same unpack + MFMA into its own accumulator, P = P_lo + 16 P_hi and t = b - 17 MAG s.
//
// HBM layout (built once at attach, lm.hip:q_repack):
//   wfrag  [tile = n/16][blk = k/128][plane][lane = 16 c + r][4 words]: word st of lane (r, c) = the plane's nibbles of
//          W[16 tile + r][128 blk + 32 st + 8 c .. +7], nibble of K-value 2 i at bits [4 i, 4 i + 4), of 2 i + 1 at bits [16 + 4 i, ..)
//   stfrag [tile][blk][row r][4] fp32: (s, t) of the block's two groups -- the weights are the MFMA's COLUMN operand, so a lane's four
//          accumulator values share one output column and one 16-byte load per block brings its scale and offset
// Results agree with the 16-bit step on the de-quantised checkpoint to the rounding of the de-quantised weights to 16 bit (this form
// does not round them at all); tests/test_lm_gpu.py compares both with the fp32 oracle.
// ------------------------------------------------------------------------------------------------
struct QFrag { const uint32_t* wfrag; const float* stfrag; };

template <typename T> struct QMagic;
template <> struct QMagic<BF16> { static constexpr uint32_t pair = 0x43004300u; static constexpr uint32_t one = 0x3f803f80u; };   // 128 + q; 1.0
template <> struct QMagic<F16> { static constexpr uint32_t pair = 0x64006400u; static constexpr uint32_t one = 0x3c003c00u; };    // 1024 + q; 1.0

// Epilogue of the transposed accumulator layout (activations are the MFMA's row operand here): lane (r, c) holds
// C[m = m0 + 16 mt + 4 c + i][n = n0 + 16 t + r], i = 0..3 -- one output column per lane, so scale / offset are per-lane scalars.
template <typename T, int MODE, int NT>
__device__ __forceinline__ void skinny_store_tr(const SkinnyArgs& a, const f32x4 (&acc)[NT][2], int n0, int m0, int split, int lane, const float* rs) {
  const int r = lane & 15, c = lane >> 4;
  const bool scaled = a.ss_in != nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = n0 + 16 * t + r;
    const bool nv = n < a.N;
    const float bs = (MODE != SK_PARTIAL && a.bias && nv) ? a.bias[n] : 0.f;
    const float wn = (MODE == SK_RESID && nv) ? a.nw[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mr = 16 * mt + 4 * c + i, m = m0 + mr;
        const bool mv = m < a.M;
        const float raw = scaled ? acc[t][mt][i] * (mv ? rs[mr] : 0.f) : acc[t][mt][i];
        float v = raw + bs;
        if (MODE == SK_SWIGLU) {                 // interleaved rows: even column = gate, odd column = up (the neighbouring lane)
          const float u = dpp_f32<0xB1>(v);      // quad_perm [1, 0, 3, 2]: every lane executes the exchange
          if (nv && mv && !(r & 1)) reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + (n >> 1)] = T::from_f32((v / (1.0f + __expf(-v))) * u);
          continue;
        }
        if (MODE == SK_RESID) {                  // x += acc; next activation = x * norm weight; the tile's sum of squares (16 lanes of a DPP row)
          float x = 0.f;
          if (nv && mv) {
            float* xp = a.xres + (int64_t)m * a.N + n;
            x = *xp + v;
            *xp = x;
            reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + n] = T::from_f32(x * wn);
          }
          float q = x * x;
          q += dpp_f32<0xB1>(q); q += dpp_f32<0x4E>(q); q += dpp_f32<0x141>(q); q += dpp_f32<0x140>(q);     // fixed butterfly: every lane of the row holds the sum
          if (r == 0 && mv) a.ss_out[(int64_t)((n0 >> 4) + t) * a.M + m] = q;
          continue;
        }
        if (!nv || !mv) continue;
        if (MODE == SK_PARTIAL) reinterpret_cast<float*>(a.out)[((int64_t)split * a.M + m) * a.N + n] = raw;
        else if (MODE == SK_OUTF32) reinterpret_cast<float*>(a.out)[(int64_t)m * a.ldo + n] = v;
        else reinterpret_cast<uint16_t*>(a.out)[(int64_t)m * a.ldo + n] = T::from_f32(v);
      }
    }
  }
}

// M16: at most 16 rows (single-sequence decode, small batches): the second 16-row MFMA half and its activation loads are skipped
// NP: nibble planes per code (1 = 4-bit, 2 = 8-bit)
template <typename T, int MODE, int NT, int NW, bool M16, int NP>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_qi(SkinnyArgs a, QFrag q) {
  __shared__ float rs[32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = blockIdx.x * NT;
  const int n0 = tile0 * 16;
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * 32;
  const int nblk = a.K >> 7;                      // 128-input blocks per row
  const int bc = nblk / (a.S * NW);               // blocks per wave
  const int b0 = (split * NW + wave) * bc;
  const int n_tiles = (a.N + 15) >> 4;
  const int r = lane & 15, c = lane >> 4;
  // activations are the MFMA's ROW operand (lane r = row m0 + r): only lanes whose row exists fetch them (dec_skinny_gemm)
  const bool av0 = m0 + r < a.M, av1 = !M16 && m0 + 16 + r < a.M;
  const uint16_t* ap0 = a.A + (int64_t)(av0 ? m0 + r : 0) * a.lda + 8 * c;
  const uint16_t* ap1 = a.A + (int64_t)(av1 ? m0 + 16 + r : 0) * a.lda + 8 * c;
  const u32x4* wp[NT];
  const f32x4* sp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tl = tile0 + t < n_tiles ? tile0 + t : n_tiles - 1;      // tiles past the end re-read the last one and are never stored
    wp[t] = reinterpret_cast<const u32x4*>(q.wfrag) + ((int64_t)tl * nblk) * (NP * 64) + lane;
    sp[t] = reinterpret_cast<const f32x4*>(q.stfrag) + ((int64_t)tl * nblk) * 16 + r;      // (s, t) of the block's two groups for column r
  }
  // codes and (scale, offset) pairs as buffer loads at 32-bit byte offsets, non-temporal (aux 2): see dec_skinny_fflat
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(q.wfrag), (short)0, (int)((unsigned)n_tiles * (unsigned)nblk * (NP * 64 * 16u)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.stfrag), (short)0, (int)((unsigned)n_tiles * (unsigned)nblk * 256u), 0x00020000);
  uint32_t wo[NT], so[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    wo[t] = (uint32_t)((const char*)wp[t] - (const char*)q.wfrag);
    so[t] = (uint32_t)((const char*)sp[t] - (const char*)q.stfrag);
  }
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const s16x8 zfrag = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
  struct Blk { u32x4 w[NT][NP]; f32x4 st[NT]; s16x8 a0[4], a1[4]; };
  auto load_blk = [&](Blk& b, int blk) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int p = 0; p < NP; ++p) b.w[t][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(wo[t] + (uint32_t)(blk * NP + p) * 1024u), 0, 2));
      b.st[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rst, (int)(so[t] + (uint32_t)blk * 256u), 0, 2));
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      b.a0[st] = zfrag;
      if (av0) b.a0[st] = *reinterpret_cast<const s16x8*>(ap0 + (int64_t)blk * 128 + 32 * st);
      if (!M16) { b.a1[st] = zfrag; if (av1) b.a1[st] = *reinterpret_cast<const s16x8*>(ap1 + (int64_t)blk * 128 + 32 * st); }
    }
  };
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  const s16x8 ones = __builtin_bit_cast(s16x8, (u32x4){QMagic<T>::one, QMagic<T>::one, QMagic<T>::one, QMagic<T>::one});
  auto unpack = [](uint32_t word) -> s16x8 {     // 8 codes -> 8 x (MAG + q) in K order
    return __builtin_bit_cast(s16x8, (u32x4){(word & 0x000f000fu) | QMagic<T>::pair, ((word >> 4) & 0x000f000fu) | QMagic<T>::pair,
                                             ((word >> 8) & 0x000f000fu) | QMagic<T>::pair, ((word >> 12) & 0x000f000fu) | QMagic<T>::pair});
  };
  auto mma_blk = [&](const Blk& b) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // A[i] = sum of the group's 64 activations of row 4 c + i
      f32x4 A0 = T::mfma16(b.a0[2 * g], ones, zero4), A1 = zero4;
      A0 = T::mfma16(b.a0[2 * g + 1], ones, A0);
      if (!M16) { A1 = T::mfma16(b.a1[2 * g], ones, zero4); A1 = T::mfma16(b.a1[2 * g + 1], ones, A1); }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4 P0[NP], P1[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const s16x8 f0 = unpack(b.w[t][p][2 * g]), f1 = unpack(b.w[t][p][2 * g + 1]);
          P0[p] = T::mfma16(b.a0[2 * g], f0, zero4);
          P0[p] = T::mfma16(b.a0[2 * g + 1], f1, P0[p]);
          if (!M16) { P1[p] = T::mfma16(b.a1[2 * g], f0, zero4); P1[p] = T::mfma16(b.a1[2 * g + 1], f1, P1[p]); }
        }
        const float sc = b.st[t][2 * g], tt = b.st[t][2 * g + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p0 = NP == 2 ? __builtin_fmaf(16.0f, P0[NP - 1][i], P0[0][i]) : P0[0][i];
          acc[t][0][i] = __builtin_fmaf(sc, p0, __builtin_fmaf(tt, A0[i], acc[t][0][i]));
          if (!M16) {
            const float p1 = NP == 2 ? __builtin_fmaf(16.0f, P1[NP - 1][i], P1[0][i]) : P1[0][i];
            acc[t][1][i] = __builtin_fmaf(sc, p1, __builtin_fmaf(tt, A1[i], acc[t][1][i]));
          }
        }
      }
    }
  };
  // ring of three register blocks (each 4 K-steps = two groups deep): two blocks of loads in flight behind the one being multiplied
  Blk k0, k1, k2;
  if (bc > 0) load_blk(k0, b0);
  if (bc > 1) load_blk(k1, b0 + 1);
  skinny_rstd_prepare(a, rs, m0, wave, lane);        // behind the first two blocks of loads: its own loads ride under the weight stream
  for (int i = 0; i < bc; i += 3) {
    if (i + 2 < bc) load_blk(k2, b0 + i + 2);
    mma_blk(k0);
    if (i + 1 >= bc) break;
    if (i + 3 < bc) load_blk(k0, b0 + i + 3);
    mma_blk(k1);
    if (i + 2 >= bc) break;
    if (i + 4 < bc) load_blk(k1, b0 + i + 4);
    mma_blk(k2);
  }
  if (!skinny_wave_reduce<NT, NW>(acc, wave, lane)) return;
  skinny_store_tr<T, MODE, NT>(a, acc, n0, m0, split, lane, rs);
}

// ------------------------------------------------------------------------------------------------
// Single-query attention against a head-major K/V cache: one workgroup per (head, clip).
// 8 lanes share a key (16 B each, fully coalesced 1 KB per wave instruction); fp32 softmax.
// n_keys = pos[clip] + 1 (self attention) or the constant T (cross attention).
// ------------------------------------------------------------------------------------------------
constexpr int DEC_MAX_KEYS = 1536;

// U = independent 1 KB loads a wave keeps in flight per trip (8 U keys); NTL = non-temporal K/V loads (the cross K/V of a step is
// 1 GB read once: keep it from displacing the step's weights and activations in L2 / MALL)
template <typename T, int U, bool NTL>
__global__ __launch_bounds__(256) void dec_attention(const uint16_t* __restrict__ q, const uint16_t* __restrict__ kc,
                                                     const uint16_t* __restrict__ vc, uint16_t* __restrict__ out,
                                                     const int32_t* __restrict__ pos_arr, int fixed_keys, int cap_keys, int H,
                                                     float scale, float* __restrict__ qk_out, const int32_t* __restrict__ head_slot,
                                                     int n_slots, int qk_ctx) {
  __shared__ float sc[DEC_MAX_KEYS];
  __shared__ float red[4][64];
  __shared__ float red2[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int D = H * 64;
  const int nk = fixed_keys > 0 ? fixed_keys : pos_arr[b] + 1;
  const int c = lane & 7, g = lane >> 3;
  const uint16_t* kb = kc + ((int64_t)b * H + h) * cap_keys * 64;
  const uint16_t* vb = vc + ((int64_t)b * H + h) * cap_keys * 64;
  float qf[8];
  {
    const s16x8 qv = *reinterpret_cast<const s16x8*>(q + (int64_t)b * D + h * 64 + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = T::to_f32((uint16_t)qv[j]);
  }
  // K / V of this (clip, head) as buffer resources: 16-byte loads at 32-bit offsets (no 64-bit address arithmetic per load), the cache
  // policy in the instruction's aux immediate (2 = non-temporal)
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kb), (short)0, cap_keys * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vb), (short)0, cap_keys * 128, 0x00020000);
  auto load_trip = [&](s16x8 (&dst)[U], const __amdgpu_buffer_rsrc_t& base, int k0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int key = k0 + 8 * u + g; key = key < nk ? key : nk - 1;
      dst[u] = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(base, key * 128 + c * 16, 0, NTL ? 2 : 0));
    }
  };
  constexpr int TRIP = 32 * U;   // keys per workgroup trip; a wave owns 8 U consecutive keys of it
  // ---- scores: the next trip's U independent 1 KB loads are issued before the current trip's dot products (U .. 2U loads in flight)
  {
    int k0 = wave * (8 * U);
    s16x8 kv[U], kn[U];
    if (k0 < nk) load_trip(kv, rk, k0);
    for (; k0 < nk; k0 += TRIP) {
      const bool more = k0 + TRIP < nk;
      if (more) load_trip(kn, rk, k0 + TRIP);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += qf[j] * T::to_f32((uint16_t)kv[u][j]);
        dot += dpp_xor1(dot);            // 8-lane all-reduce on DPP (the lanes of a key): no LDS-crossbar round trips in the K loop
        dot += dpp_xor2(dot);
        dot += dpp_half_mirror(dot);
        const int key = k0 + 8 * u + g;
        if (c == 0 && key < nk) sc[key] = dot * scale;
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) kv[u] = kn[u];
      }
    }
  }
  // the first V trip does not depend on the probabilities: its loads fly during the softmax
  s16x8 vv[U], vn[U];
  if (wave * (8 * U) < nk) load_trip(vv, rv, wave * (8 * U));
  __syncthreads();
  // word-timestamp alignment (WhisperTiming.swift:605-640): keep the pre-softmax scores of the alignment heads, row = decoder position
  if (qk_out && head_slot[h] >= 0) {
    float* dst = qk_out + (((int64_t)b * n_slots + head_slot[h]) * qk_ctx + pos_arr[b]) * nk;
    for (int i = tid; i < nk; i += 256) dst[i] = sc[i];
  }
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
  for (int k0 = wave * (8 * U); k0 < nk; k0 += TRIP) {
    const bool more = k0 + TRIP < nk;
    if (more) load_trip(vn, rv, k0 + TRIP);
    float pw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = k0 + 8 * u + g;
      pw[u] = key < nk ? sc[key] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += pw[u] * T::to_f32((uint16_t)vv[u][j]);
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) vv[u] = vn[u];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] += dpp_xor8(acc[j]);
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
    out[afrag_index(b, h * 64 + tid, D)] = T::from_f32(o);   // fragment order: the A operand of the output projection
  }
}

// ------------------------------------------------------------------------------------------------
// Decode head: logit rules + argmax + log-prob bookkeeping (WhisperDecoding.swift:158-169,186-358), one workgroup
// per clip, all reductions in fp32; each block advances its own clip's position.
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
  const float* uniforms;        // [B][n_ctx] explicit RNG for temperature > 0 (one value per generated token)
  DecClip clip;
};

constexpr int HEAD_NPT = 52;   // logits per thread held in registers: V <= 1024 * 52
struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax amax(ArgMax a, ArgMax b) {   // larger value wins, lowest index wins ties (MLX argMax)
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax wave_amax(ArgMax a) {
  // rows of 16 on DPP, then the four row winners as scalars (amax is associative and commutative: any order gives the same pair)
#define AMAX_STEP(CTRL) { const ArgMax b{dpp_f32<CTRL>(a.v), dpp_i32<CTRL>(a.i)}; a = amax(a, b); }
  AMAX_STEP(0xB1) AMAX_STEP(0x4E) AMAX_STEP(0x141) AMAX_STEP(0x140)
#undef AMAX_STEP
  ArgMax r = ArgMax{lane_f32(a.v, 0), __builtin_amdgcn_readlane(a.i, 0)};
  r = amax(r, ArgMax{lane_f32(a.v, 16), __builtin_amdgcn_readlane(a.i, 16)});
  r = amax(r, ArgMax{lane_f32(a.v, 32), __builtin_amdgcn_readlane(a.i, 32)});
  r = amax(r, ArgMax{lane_f32(a.v, 48), __builtin_amdgcn_readlane(a.i, 48)});
  return r;
}

// One workgroup per clip.  The clip's logits are read once into registers; pass 1 = every max / argmax, pass 2 = every
// exp-sum.  The timestamp heuristic (:299-322) decides between two precomputed candidates: A = rules only, B = rules +
// "text suppressed".  ts_lse = (max_ts - lse) + log(sum_ts exp(x - max_ts)).  temperature > 0: inverse-CDF sampling of
// softmax(filtered / T) with the caller's uniform (sampleFromDistribution, :395-410: first index whose cumsum >= r).
// Each block advances its own clip's position; a finished clip stops advancing.
__global__ __launch_bounds__(1024) void dec_head(HeadBufs hb, DecodeParams p) {
  __shared__ float shf[16][4];
  __shared__ float sha[16][2];
  __shared__ int shai[16][2];
  __shared__ float seg[HEAD_NPT * 16];
  __shared__ int s_pick[2];
  __shared__ float s_val[2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V;
  const float* lg = hb.logits + (int64_t)b * V;
  const int pos = hb.clip.pos[b];
  const int n_initial = hb.clip.n_init[b], sot_index = hb.clip.sot_idx[b];
  const float temperature = hb.clip.temp[b];
  const int cur_len = pos + 1;
  int32_t* toks = hb.tokens + (int64_t)b * p.n_ctx;
  const bool generating = cur_len >= n_initial;
  if (generating && hb.finished[b]) return;           // done: the clip idles at its last position
  if (!generating && pos != sot_index) {              // forced token and no probe wanted: just advance
    __syncthreads();
    if (tid == 0) hb.clip.pos[b] = pos + 1;
    return;
  }
  const bool decide = generating;
  const int num_gen = cur_len - n_initial;            // == loop iteration of the reference
  const int tsb = p.timestamp_begin;

  // ---- rule state (WhisperDecoding.swift:221-292)
  bool sup_ts_all = false, sup_text_below_eot = false, sup_below_tsb = false, heuristic = false;
  int ts_floor = 0;                                   // suppress tsb <= idx < ts_floor
  int max_first = V;                                  // suppress idx > max_first (first token only)
  if (decide && p.timestamps) {
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
    heuristic = num_gen > 0;
  }
  const int nw = (V + 31) / 32;
  const uint32_t* bits = hb.suppress + (num_gen == 0 ? nw : 0);
  auto maskedA = [&](int i) -> bool {
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

  // ---- the clip's logits are read ONCE: 52 registers per thread (V <= 53248), all loads in flight together
  float x[HEAD_NPT];
#pragma unroll
  for (int u = 0; u < HEAD_NPT; ++u) { const int i = tid + 1024 * u; x[u] = i < V ? lg[i] : -INFINITY; }
  // ---- pass 1: maxima
  float mx_text = -INFINITY, mx_ts = -INFINITY;
  ArgMax bA{-INFINITY, 0x7fffffff}, bB{-INFINITY, 0x7fffffff};
  unsigned long long okA = 0ull;                      // bit u: element u is unmasked under the rules (hypothesis A)
#pragma unroll
  for (int u = 0; u < HEAD_NPT; ++u) {
    const int i = tid + 1024 * u;
    if (i >= V) continue;
    if (i >= tsb) mx_ts = fmaxf(mx_ts, x[u]); else mx_text = fmaxf(mx_text, x[u]);
    if (decide && !maskedA(i)) {
      okA |= 1ull << u;
      bA = amax(bA, ArgMax{x[u], i});
      if (i >= tsb) bB = amax(bB, ArgMax{x[u], i});
    }
  }
  mx_text = wave_max(mx_text); mx_ts = wave_max(mx_ts);
  bA = wave_amax(bA); bB = wave_amax(bB);
  if (lane == 0) { shf[wave][0] = mx_text; shf[wave][1] = mx_ts; sha[wave][0] = bA.v; shai[wave][0] = bA.i; sha[wave][1] = bB.v; shai[wave][1] = bB.i; }
  __syncthreads();
  mx_text = shf[0][0]; mx_ts = shf[0][1]; bA = ArgMax{sha[0][0], shai[0][0]}; bB = ArgMax{sha[0][1], shai[0][1]};
  for (int w2 = 1; w2 < 16; ++w2) {
    mx_text = fmaxf(mx_text, shf[w2][0]); mx_ts = fmaxf(mx_ts, shf[w2][1]);
    bA = amax(bA, ArgMax{sha[w2][0], shai[w2][0]}); bB = amax(bB, ArgMax{sha[w2][1], shai[w2][1]});
  }
  const float mx_all = fmaxf(mx_text, mx_ts);
  __syncthreads();

  // ---- pass 2: exp-sums (registers)
  float s_all = 0.f, s_ts = 0.f, fA = 0.f, fB = 0.f;
#pragma unroll
  for (int u = 0; u < HEAD_NPT; ++u) {
    const int i = tid + 1024 * u;
    if (i >= V) continue;
    s_all += __expf(x[u] - mx_all);
    if (heuristic && i >= tsb) s_ts += __expf(x[u] - mx_ts);
    if ((okA >> u) & 1ull) {
      fA += __expf(x[u] - bA.v);
      if (heuristic && i >= tsb) fB += __expf(x[u] - bB.v);
    }
  }
  s_all = wave_sum(s_all); s_ts = wave_sum(s_ts); fA = wave_sum(fA); fB = wave_sum(fB);
  if (lane == 0) { shf[wave][0] = s_all; shf[wave][1] = s_ts; shf[wave][2] = fA; shf[wave][3] = fB; }
  __syncthreads();
  s_all = s_ts = fA = fB = 0.f;
  for (int w2 = 0; w2 < 16; ++w2) { s_all += shf[w2][0]; s_ts += shf[w2][1]; fA += shf[w2][2]; fB += shf[w2][3]; }
  const float lse = mx_all + __logf(s_all);
  if (pos == sot_index && tid == 0) hb.no_speech[b] = __expf(lg[p.no_speech] - lse);   // softmax(logits[sot])[no_speech] (:158-169)
  if (!decide) {
    if (tid == 0) hb.clip.pos[b] = pos + 1;
    return;
  }
  bool useB = false;
  if (heuristic) {
    const float ts_lse = (mx_ts - lse) + __logf(s_ts);
    const float max_text = mx_text - lse;
    useB = ts_lse > max_text;                         // force a timestamp: text tokens suppressed too
  }
  const ArgMax best = useB ? bB : bA;
  const float fsum = useB ? fB : fA;
  // Everything masked: reachable in the reference when the raw-logit timestamp heuristic fires right after a
  // timestamp pair rule; MLX argMax of an all -inf vector is index 0 and log(softmax) is NaN.  Mirror that.
  const bool all_masked = best.i == 0x7fffffff;
  int next = all_masked ? 0 : best.i;

  if (temperature > 0.0f && !all_masked) {
    // ---- sampling: p_i ~ exp((x_i - max) / T) over the kept set, first index whose cumulative sum >= r * total
    const float inv_t = 1.0f / temperature;
    auto weight = [&](int u) -> float {
      const int i = tid + 1024 * u;
      const bool kept = ((okA >> u) & 1ull) && (!useB || i >= tsb);
      return kept ? __expf((x[u] - best.v) * inv_t) : 0.f;
    };
#pragma unroll
    for (int u = 0; u < HEAD_NPT; ++u) {
      const float ssum = wave_sum(weight(u));         // segment (u, wave) covers indices [1024u + 64*wave, +64): index-ordered
      if (lane == 0) seg[u * 16 + wave] = ssum;
    }
    __syncthreads();
    if (tid == 0) {
      float total = 0.f;
      for (int k = 0; k < HEAD_NPT * 16; ++k) total += seg[k];
      const float r = hb.uniforms[(int64_t)b * p.n_ctx + num_gen];
      const float goal = r * total;
      float cum = 0.f; int sel = -1;
      for (int k = 0; k < HEAD_NPT * 16; ++k) {
        if (seg[k] > 0.f && cum + seg[k] >= goal) { sel = k; break; }
        cum += seg[k];
      }
      if (sel < 0) { for (int k = HEAD_NPT * 16 - 1; k >= 0; --k) if (seg[k] > 0.f) { sel = k; break; } cum -= 0.f; }
      s_pick[0] = sel; s_val[0] = cum; s_val[1] = goal;
    }
    __syncthreads();
    const int sel = s_pick[0];
    if (sel >= 0 && wave == (sel & 15)) {
      const int u_sel = sel >> 4;
      float mine = 0.f;
#pragma unroll
      for (int u = 0; u < HEAD_NPT; ++u) if (u == u_sel) mine = weight(u);
      float incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      const unsigned long long hit = __ballot(mine > 0.f && s_val[0] + incl >= s_val[1]);
      const unsigned long long any = __ballot(mine > 0.f);
      int ln = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)any);   // rounding corner: last kept lane of the segment
      if (lane == 0) s_pick[1] = 1024 * u_sel + 64 * wave + ln;
    }
    __syncthreads();
    if (sel >= 0) next = s_pick[1];
  }
  if (tid != 0) return;
  if (next != p.eot) {                                // EOT excluded from avg_logprob (:345-350)
    hb.sum_logprob[b] += all_masked ? __int_as_float(0x7fc00000) : (lg[next] - best.v) - __logf(fsum);
    hb.n_logprob[b] += 1;
  }
  toks[cur_len] = next;
  hb.n_gen[b] = num_gen + 1;
  if (next > tsb) hb.last_ts[b] = next;               // strict '>' (:254-256)
  int cap = p.max_tokens - n_initial;
  if (p.max_new_tokens > 0 && p.max_new_tokens < cap) cap = p.max_new_tokens;
  if (next == p.eot || num_gen + 1 >= cap) hb.finished[b] = 1;
  else hb.clip.pos[b] = pos + 1;
}

// ------------------------------------------------------------------------------------------------
// Greedy (temperature 0) head split over HEAD_SPLIT workgroups per clip: one workgroup per clip left 224 of the 256 CUs idle while
// 32 of them chewed 52 k logits each (46 us per step).  Phase 1 reduces a 1/HEAD_SPLIT slice of the vocabulary to a record of
// (max, argmax, exp-sum) pairs taken relative to the slice's own maxima; phase 2 rescales and merges the records in slice order and
// applies exactly the decision logic of dec_head.  Sampling (temperature > 0) keeps the single-workgroup kernel.
// ------------------------------------------------------------------------------------------------
constexpr int HEAD_SPLIT = 32;   // <= 64: dec_head_final merges one record per lane
constexpr int HEAD_NPT2 = 7;       // logits per thread: 256 threads * 7 * 32 slices >= V
struct HeadRule {
  bool active, decide, heuristic, sup_ts_all, sup_text_below_eot, sup_below_tsb;
  int ts_floor, max_first, num_gen, cur_len;
};
__device__ __forceinline__ HeadRule head_rule(const HeadBufs& hb, const DecodeParams& p, int b) {
  HeadRule r{};
  const int pos = hb.clip.pos[b], n_initial = hb.clip.n_init[b], sot_index = hb.clip.sot_idx[b];
  r.cur_len = pos + 1;
  const bool generating = r.cur_len >= n_initial;
  r.active = !(generating && hb.finished[b]) && (generating || pos == sot_index);
  r.decide = generating;
  r.num_gen = r.cur_len - n_initial;
  r.max_first = p.V;
  if (r.active && r.decide && p.timestamps) {
    const int32_t* toks = hb.tokens + (int64_t)b * p.n_ctx;
    const int tsb = p.timestamp_begin;
    const int last = toks[r.cur_len - 1];
    const bool last_was_ts = r.num_gen >= 1 && last >= tsb;
    const bool penult_was_ts = r.num_gen < 2 || toks[r.cur_len - 2] >= tsb;
    if (last_was_ts) { if (penult_was_ts) r.sup_ts_all = true; else r.sup_text_below_eot = true; }
    const int lt = hb.last_ts[b];
    if (lt > 0) r.ts_floor = penult_was_ts ? lt + 1 : lt;
    if (r.num_gen == 0) {
      r.sup_below_tsb = true;
      const int last_allowed = tsb + p.max_initial_ts;
      if (last_allowed < p.V) r.max_first = last_allowed;
    }
    r.heuristic = r.num_gen > 0;
  }
  return r;
}

struct HeadPart { float mx_text, mx_ts, s_all, s_ts, bAv, bBv, fA, fB; int bAi, bBi; };   // sums are relative to this slice's maxima

__global__ __launch_bounds__(256) void dec_head_partial(HeadBufs hb, DecodeParams p, HeadPart* __restrict__ parts) {
  __shared__ float shf[4][4];
  __shared__ float sha[4][2];
  __shared__ int shai[4][2];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = p.V, tsb = p.timestamp_begin;
  const float* lg = hb.logits + (int64_t)b * V;
  // the slice's logits do not depend on the clip's rule state: their loads go out first, so the rule's chain of small dependent
  // loads (position -> generated count -> last tokens) runs in their shadow instead of ahead of them
  const int base = k * (256 * HEAD_NPT2);
  float x[HEAD_NPT2];
#pragma unroll
  for (int u = 0; u < HEAD_NPT2; ++u) { const int i = base + tid + 256 * u; x[u] = i < V ? lg[i] : -INFINITY; }
  const HeadRule r = head_rule(hb, p, b);
  if (!r.active) return;
  const int nw = (V + 31) / 32;
  const uint32_t* bits = hb.suppress + (r.num_gen == 0 ? nw : 0);
  auto maskedA = [&](int i) -> bool {
    if ((bits[i >> 5] >> (i & 31)) & 1u) return true;
    if (p.timestamps) {
      if (i == p.no_timestamps) return true;
      if (r.sup_ts_all && i >= tsb) return true;
      if (r.sup_text_below_eot && i < p.eot) return true;
      if (i >= tsb && i < r.ts_floor) return true;
      if (r.sup_below_tsb && i < tsb) return true;
      if (i > r.max_first) return true;
    }
    return false;
  };
  float mx_text = -INFINITY, mx_ts = -INFINITY;
  ArgMax bA{-INFINITY, 0x7fffffff}, bB{-INFINITY, 0x7fffffff};
  unsigned okA = 0u;
#pragma unroll
  for (int u = 0; u < HEAD_NPT2; ++u) {
    const int i = base + tid + 256 * u;
    if (i >= V) continue;
    if (i >= tsb) mx_ts = fmaxf(mx_ts, x[u]); else mx_text = fmaxf(mx_text, x[u]);
    if (r.decide && !maskedA(i)) {
      okA |= 1u << u;
      bA = amax(bA, ArgMax{x[u], i});
      if (i >= tsb) bB = amax(bB, ArgMax{x[u], i});
    }
  }
  mx_text = wave_max(mx_text); mx_ts = wave_max(mx_ts);
  bA = wave_amax(bA); bB = wave_amax(bB);
  if (lane == 0) { shf[wave][0] = mx_text; shf[wave][1] = mx_ts; sha[wave][0] = bA.v; shai[wave][0] = bA.i; sha[wave][1] = bB.v; shai[wave][1] = bB.i; }
  __syncthreads();
  mx_text = shf[0][0]; mx_ts = shf[0][1]; bA = ArgMax{sha[0][0], shai[0][0]}; bB = ArgMax{sha[0][1], shai[0][1]};
  for (int w2 = 1; w2 < 4; ++w2) {
    mx_text = fmaxf(mx_text, shf[w2][0]); mx_ts = fmaxf(mx_ts, shf[w2][1]);
    bA = amax(bA, ArgMax{sha[w2][0], shai[w2][0]}); bB = amax(bB, ArgMax{sha[w2][1], shai[w2][1]});
  }
  const float mx_all = fmaxf(mx_text, mx_ts);
  __syncthreads();
  float s_all = 0.f, s_ts = 0.f, fA = 0.f, fB = 0.f;
#pragma unroll
  for (int u = 0; u < HEAD_NPT2; ++u) {
    const int i = base + tid + 256 * u;
    if (i >= V) continue;
    s_all += __expf(x[u] - mx_all);
    if (r.heuristic && i >= tsb) s_ts += __expf(x[u] - mx_ts);
    if ((okA >> u) & 1u) {
      fA += __expf(x[u] - bA.v);
      if (r.heuristic && i >= tsb) fB += __expf(x[u] - bB.v);
    }
  }
  s_all = wave_sum(s_all); s_ts = wave_sum(s_ts); fA = wave_sum(fA); fB = wave_sum(fB);
  if (lane == 0) { shf[wave][0] = s_all; shf[wave][1] = s_ts; shf[wave][2] = fA; shf[wave][3] = fB; }
  __syncthreads();
  if (tid == 0) {
    s_all = s_ts = fA = fB = 0.f;
    for (int w2 = 0; w2 < 4; ++w2) { s_all += shf[w2][0]; s_ts += shf[w2][1]; fA += shf[w2][2]; fB += shf[w2][3]; }
    parts[b * HEAD_SPLIT + k] = HeadPart{mx_text, mx_ts, s_all, s_ts, bA.v, bB.v, fA, fB, bA.i, bB.i};
  }
}

// merge the slice records in slice order (deterministic), decide, update the clip's state (same tail as dec_head); then, if the clip
// advanced, embed the token of the NEW position and apply the first LayerNorm of the next step (replaces that step's dec_embed_ln launch)
struct NextEmbed { const uint16_t* emb; const float* pos_emb; const float* gamma; const float* beta; float* x; uint16_t* h; int D; };

template <typename T>
__global__ __launch_bounds__(256) void dec_head_final(HeadBufs hb, DecodeParams p, const HeadPart* __restrict__ parts, NextEmbed ne) {
  __shared__ int s_newpos;
  __shared__ float sh[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  // wave 0 merges the clip's slice records, one per lane: maxima by butterfly, then the exp-sums rescaled to the merged maxima
  // (fixed butterfly order: deterministic)
  float mx_text = -INFINITY, mx_ts = -INFINITY, s_all = 0.f, s_ts = 0.f, fA = 0.f, fB = 0.f;
  ArgMax bA{-INFINITY, 0x7fffffff}, bB{-INFINITY, 0x7fffffff};
  const HeadRule r = head_rule(hb, p, b);
  if (tid < 64 && r.active) {
    HeadPart me{-INFINITY, -INFINITY, 0.f, 0.f, -INFINITY, -INFINITY, 0.f, 0.f, 0x7fffffff, 0x7fffffff};
    if (tid < HEAD_SPLIT) me = parts[b * HEAD_SPLIT + tid];
    mx_text = wave_max(me.mx_text); mx_ts = wave_max(me.mx_ts);
    bA = wave_amax(ArgMax{me.bAv, me.bAi}); bB = wave_amax(ArgMax{me.bBv, me.bBi});
    const float mx_all = fmaxf(mx_text, mx_ts);
    const float ml = fmaxf(me.mx_text, me.mx_ts);
    s_all = wave_sum(ml > -INFINITY ? me.s_all * __expf(ml - mx_all) : 0.f);
    s_ts = wave_sum(me.mx_ts > -INFINITY ? me.s_ts * __expf(me.mx_ts - mx_ts) : 0.f);
    fA = wave_sum(me.bAv > -INFINITY ? me.fA * __expf(me.bAv - bA.v) : 0.f);
    fB = wave_sum(me.bBv > -INFINITY ? me.fB * __expf(me.bBv - bB.v) : 0.f);
  }
  if (tid == 0) {
    int newpos = -1;
    const int pos = hb.clip.pos[b];
    const int n_initial = hb.clip.n_init[b], sot_index = hb.clip.sot_idx[b];
    if (!r.active) {
      if (r.cur_len < n_initial) newpos = pos + 1;    // forced token, no probe: just advance (a finished clip idles)
    } else {
      const int V = p.V, tsb = p.timestamp_begin;
      const float* lg = hb.logits + (int64_t)b * V;
      int32_t* toks = hb.tokens + (int64_t)b * p.n_ctx;
      const float mx_all = fmaxf(mx_text, mx_ts);
      const float lse = mx_all + __logf(s_all);
      if (pos == sot_index) hb.no_speech[b] = __expf(lg[p.no_speech] - lse);
      if (!r.decide) newpos = pos + 1;
      else {
        bool useB = false;
        if (r.heuristic) {
          const float ts_lse = (mx_ts - lse) + __logf(s_ts);
          const float max_text = mx_text - lse;
          useB = ts_lse > max_text;
        }
        const ArgMax best = useB ? bB : bA;
        const float fsum = useB ? fB : fA;
        const bool all_masked = best.i == 0x7fffffff;
        const int next = all_masked ? 0 : best.i;
        if (next != p.eot) {
          hb.sum_logprob[b] += all_masked ? __int_as_float(0x7fc00000) : (lg[next] - best.v) - __logf(fsum);
          hb.n_logprob[b] += 1;
        }
        toks[r.cur_len] = next;
        hb.n_gen[b] = r.num_gen + 1;
        if (next > tsb) hb.last_ts[b] = next;
        int cap = p.max_tokens - n_initial;
        if (p.max_new_tokens > 0 && p.max_new_tokens < cap) cap = p.max_new_tokens;
        if (next == p.eot || r.num_gen + 1 >= cap) hb.finished[b] = 1;
        else newpos = pos + 1;
      }
    }
    if (newpos >= 0) hb.clip.pos[b] = newpos;
    s_newpos = newpos;
  }
  __syncthreads();
  const int np_ = s_newpos;
  if (np_ < 0 || np_ >= p.n_ctx) return;
  // ---- x[b] = E[token[b][np]] + P[np];  h[b] = LN(x[b])   (the next step's TextDecoder.swift:67 + first attn_ln)
  const int D = ne.D, nv = D >> 2;
  const int tok = hb.tokens[(int64_t)b * p.n_ctx + np_];
  const uint16_t* e = ne.emb + (int64_t)tok * D;
  const float* pe = ne.pos_emb + (int64_t)np_ * D;
  float* xr = ne.x + (int64_t)b * D;
  f32x4 v[ROW_NV];
#pragma unroll
  for (int i = 0; i < ROW_NV; ++i) {
    const int c = tid + 256 * i;
    if (c < nv) {
      const s16x4 ev = *reinterpret_cast<const s16x4*>(e + 4 * c);
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pe + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = T::to_f32((uint16_t)ev[j]) + pv[j];
      *reinterpret_cast<f32x4*>(xr + 4 * c) = v[i];
    } else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  row_layernorm_store<T>(v, nv, D, ne.gamma, ne.beta, ne.h, b, sh);
}

// compact outputs: generated tokens with EOT (and anything after) stripped; avg_logprob
__global__ void dec_finalize(const int32_t* __restrict__ tokens, const int32_t* __restrict__ n_gen,
                             const float* __restrict__ sum_lp, const int32_t* __restrict__ n_lp, int32_t* __restrict__ out_tokens,
                             int32_t* __restrict__ out_n, float* __restrict__ out_avg, const int32_t* __restrict__ n_init, DecodeParams p) {
  const int b = blockIdx.x;
  const int32_t* t = tokens + (int64_t)b * p.n_ctx + n_init[b];
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
    hipLaunchKernelGGL(dec_embed_ln<F16>, dim3(w->cur_B), dim3(256), 0, s, w->tokens, (const uint16_t*)w->tok_emb, w->dec_pos, ln.g, ln.b, w->dx, (uint16_t*)w->dh, w->clip.pos, D, w->dims.n_text_ctx);
  else
    hipLaunchKernelGGL(dec_embed_ln<BF16>, dim3(w->cur_B), dim3(256), 0, s, w->tokens, (const uint16_t*)w->tok_emb, w->dec_pos, ln.g, ln.b, w->dx, (uint16_t*)w->dh, w->clip.pos, D, w->dims.n_text_ctx);
  return 0;
}

int dec_launch_reduce_ln(mia_whisper* w, int S, const float* bias, const LNW& ln, hipStream_t s) {
  const int D = w->dims.n_text_state;
  if (w->dtype == MIA_F16)
    hipLaunchKernelGGL(dec_reduce_ln<F16>, dim3(w->cur_B), dim3(256), 0, s, w->partial, S, w->cur_B, bias, ln.g, ln.b, w->dx, (uint16_t*)w->dh, D);
  else
    hipLaunchKernelGGL(dec_reduce_ln<BF16>, dim3(w->cur_B), dim3(256), 0, s, w->partial, S, w->cur_B, bias, ln.g, ln.b, w->dx, (uint16_t*)w->dh, D);
  return 0;
}

// short-K-slice form (dec_skinny_flat): K per split = NW x 32 x NSTEP for one of the instantiated NSTEP
// short-K-slice form (dec_skinny_flat) of the row-major kernel, taken only where it keeps the ring kernel's 4-wave split of K:
// the LM step's packed-weight twin (skinny_gemm_q4) must sum in the same order as its 16-bit form
template <typename T, int MODE>
static bool skinny_flat_try(const SkinnyArgs& a, hipStream_t s) {
  const int tiles = (a.N + 15) / 16, zb = (a.M + 31) / 32;
  const int64_t wgs = (int64_t)tiles * a.S * zb;
  if (a.K % (128 * a.S) != 0 || wgs > 1024 || a.K / a.S < 512) return false;
  const int per_split = a.K / a.S;
  const dim3 grid(tiles, a.S, zb);
#define FLAT(NSTEP_)                                                                                                  \
  if (per_split == 4 * 32 * NSTEP_) {                                                                                 \
    hipLaunchKernelGGL((dec_skinny_flat<T, MODE, 1, NSTEP_, 4>), grid, dim3(256), 0, s, a);                            \
    return true;                                                                                                      \
  }
  FLAT(10) FLAT(5) FLAT(8) FLAT(6) FLAT(4)
#undef FLAT
  return false;
}

template <typename T>
static void skinny_launch_t(const SkinnyArgs& a, int mode, hipStream_t s) {
  switch (mode) {
    case SK_OUTF32: break;   // bandwidth-bound: the ring kernel (measured: flat forms are 11-14 us slower on the 51866-wide head)
    case SK_OUT16: if (skinny_flat_try<T, SK_OUT16>(a, s)) return; break;
    case SK_PARTIAL: if (skinny_flat_try<T, SK_PARTIAL>(a, s)) return; break;
    case SK_SWIGLU: if (skinny_flat_try<T, SK_SWIGLU>(a, s)) return; break;
    case SK_RESID: break;    // N / 16 workgroups only: always the ring kernel with K over 4 / 8 / 16 waves (below)
    default: if (skinny_flat_try<T, SK_QKV>(a, s)) return; break;
  }
  if (mode == SK_OUTF32) {   // the vocabulary-wide logits GEMM: 64 columns per wave
    dim3 grid((a.N + 63) / 64, a.S, (a.M + 31) / 32);
    hipLaunchKernelGGL((dec_skinny_gemm<T, SK_OUTF32, 4, 2, 1>), grid, dim3(64), 0, s, a);
    return;
  }
  dim3 grid((a.N + 15) / 16, a.S, (a.M + 31) / 32);
  // few workgroups and a long K per wave -> split K over 4 waves of the workgroup (K per wave stays a multiple of 32)
  const bool wide = (int64_t)grid.x * grid.y * grid.z <= 1024 && a.K % (128 * a.S) == 0 && a.K / a.S >= 512;
#define SK_LAUNCH(MODE_)                                                                                              \
  do {                                                                                                                \
    if (wide) hipLaunchKernelGGL((dec_skinny_gemm<T, MODE_, 1, 2, 4>), grid, dim3(256), 0, s, a);                      \
    else hipLaunchKernelGGL((dec_skinny_gemm<T, MODE_, 1, 4, 1>), grid, dim3(64), 0, s, a);                            \
  } while (0)
  switch (mode) {
    case SK_OUT16: SK_LAUNCH(SK_OUT16); break;
    case SK_PARTIAL: SK_LAUNCH(SK_PARTIAL); break;
    case SK_SWIGLU: SK_LAUNCH(SK_SWIGLU); break;
    case SK_RESID:
      // no cross-workgroup split here (the epilogue owns the residual row slice): N / 16 workgroups only, so K goes over 8 waves
      if (a.K % 1024 == 0 && a.K >= 8192) hipLaunchKernelGGL((dec_skinny_gemm<T, SK_RESID, 1, 2, 16>), grid, dim3(1024), 0, s, a);
      else if (a.K % 256 == 0 && a.K >= 2048) hipLaunchKernelGGL((dec_skinny_gemm<T, SK_RESID, 1, 2, 8>), grid, dim3(512), 0, s, a);
      else SK_LAUNCH(SK_RESID);
      break;
    default: SK_LAUNCH(SK_QKV); break;
  }
#undef SK_LAUNCH
}

int skinny_gemm_launch(const SkinnyArgs& a, int mode, int dtype, hipStream_t s) {
  if (a.K % (32 * a.S) != 0 || a.lda % 8 != 0) return -1;
  if (mode == SK_SWIGLU && (a.N & 3)) return -1;
  if (mode == SK_RESID && (a.S != 1 || (a.N & 15) || (a.ldo & 3) || !a.xres || !a.nw || !a.ss_out)) return -1;
  if (a.ss_in && (a.ss_tiles <= 0 || a.ss_dim <= 0)) return -1;
  if (dtype == MIA_F16) skinny_launch_t<F16>(a, mode, s); else skinny_launch_t<BF16>(a, mode, s);
  return 0;
}


template <typename T, bool M16, int NP>
static void skinny_qi_launch_m(const SkinnyArgs& a, const QFrag& q, int mode, hipStream_t s) {
  const int tiles = (a.N + 15) / 16, nblk = a.K / 128;
  const int zb = (a.M + 31) / 32;
  const int per_split = nblk / a.S;               // 128-input blocks per cross-workgroup split
  // 4 tiles per wave only when one tile per wave would put more than ~16 waves on every SIMD anyway (the vocabulary-wide head): the
  // activation fragments and group sums are then reused four times.  Otherwise one tile per wave and as many waves per workgroup
  // (1..4, splitting the K range) as keep 2+ blocks per wave -- the kernel hides its memory latency by occupancy.
  // ... and whenever more than 16 rows are multiplied: every wave then loads 32 activation rows per K-step (8 KB per 128-input block against
  // 1 KB of codes), which four tiles share (32 sequences side by side, Orpheus-3B: 4 900 tokens/s with one tile per wave, 11 000 with four)
  const bool nt4 = (int64_t)tiles * a.S * zb >= 16384 || (a.M > 16 && (int64_t)((tiles + 3) / 4) * a.S * zb >= 192);
  int nw = 1;
  for (int cand : {4, 3, 2}) if (per_split % cand == 0 && (per_split / cand >= 2 || cand == 2)) { nw = cand; break; }
  // the vocabulary-wide head at <= 16 rows: 4 tiles per ONE-wave workgroup, the whole K range in the wave (no LDS reduction, a quarter
  // of the workgroups) -- Orpheus-3B, V 156 940: 75.4 us as 9 809 four-wave workgroups, 72.6 as one-wave ones, 67.8 in this form
  // (4 tiles x 4 waves: 107.6)
  const bool head41 = mode == SK_OUTF32 && tiles >= 4096 && a.M <= 16;
#define QI_GO(MODE_, NT_, NW_) hipLaunchKernelGGL((skinny_gemm_qi<T, MODE_, NT_, NW_, M16, NP>), dim3((tiles + NT_ - 1) / NT_, a.S, zb), dim3(64 * NW_), 0, s, a, q)
#define QI_LAUNCH(MODE_)                                                                                   \
  do {                                                                                                     \
    if (head41) QI_GO(MODE_, 4, 1);                                                                        \
    else if (nt4) { if (per_split % 4 == 0) QI_GO(MODE_, 4, 4); else QI_GO(MODE_, 4, 1); }                  \
    else if (nw == 4) QI_GO(MODE_, 1, 4); else if (nw == 3) QI_GO(MODE_, 1, 3);                             \
    else if (nw == 2) QI_GO(MODE_, 1, 2); else QI_GO(MODE_, 1, 1);                                          \
  } while (0)
  switch (mode) {
    case SK_OUTF32: QI_LAUNCH(SK_OUTF32); break;
    case SK_OUT16: QI_LAUNCH(SK_OUT16); break;
    case SK_SWIGLU: QI_LAUNCH(SK_SWIGLU); break;
    case SK_RESID:
      if (!nt4 && per_split % 16 == 0 && per_split >= 64) QI_GO(SK_RESID, 1, 16);            // N / 16 workgroups only: K over 8 or 16 waves
      else if (!nt4 && per_split % 8 == 0) QI_GO(SK_RESID, 1, 8);
      else QI_LAUNCH(SK_RESID);
      break;
    default: QI_LAUNCH(SK_PARTIAL); break;
  }
#undef QI_LAUNCH
#undef QI_GO
}

template <typename T>
static void skinny_qi_launch_t(const SkinnyArgs& a, const QFrag& q, int bits, int mode, hipStream_t s) {
  if (bits == 8) { if (a.M <= 16) skinny_qi_launch_m<T, true, 2>(a, q, mode, s); else skinny_qi_launch_m<T, false, 2>(a, q, mode, s); }
  else { if (a.M <= 16) skinny_qi_launch_m<T, true, 1>(a, q, mode, s); else skinny_qi_launch_m<T, false, 1>(a, q, mode, s); }
}

// quantised form of skinny_gemm_launch: a.W is ignored, the weights come from the fragment-ordered arrays (see skinny_gemm_qi)
int skinny_gemm_q_launch(const SkinnyArgs& a, const uint32_t* wfrag, const float* stfrag, int bits, int mode, int dtype, hipStream_t s) {
  if (a.K % (128 * a.S) != 0 || a.lda % 8 != 0 || !wfrag || !stfrag) return -1;
  if (bits != 4 && bits != 8) return -1;
  if (mode != SK_OUT16 && mode != SK_OUTF32 && mode != SK_PARTIAL && mode != SK_SWIGLU && mode != SK_RESID) return -1;
  if (mode == SK_SWIGLU && (a.N & 3)) return -1;
  if (mode == SK_RESID && (a.S != 1 || (a.N & 15) || !a.xres || !a.nw || !a.ss_out)) return -1;
  if (a.ss_in && (a.ss_tiles <= 0 || a.ss_dim <= 0)) return -1;
  const QFrag q{wfrag, stfrag};
  if (dtype == MIA_F16) skinny_qi_launch_t<F16>(a, q, bits, mode, s); else skinny_qi_launch_t<BF16>(a, q, bits, mode, s);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the Whisper step: fragment-order operands (a.A: activation fragments, a.W: LinearW::wf)
template <typename T, int MODE>
static int skinny_frag_launch_m(const SkinnyArgs& a, hipStream_t s) {
  const int tiles = (a.N + 15) / 16, zb = (a.M + 31) / 32, ksteps = a.K / 32;
  if constexpr (MODE == SK_OUTF32) {   // the vocabulary-wide logits GEMM (bandwidth-bound): 64 columns per one-wave workgroup, ring pipeline
    hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 4, 2, 1>), dim3((tiles + 3) / 4, a.S, zb), dim3(64), 0, s, a);
    return 0;
  } else {
    // K of a split is divided over the most waves of {4, 2, 1} that take whole K-steps (summed through LDS in wave order)
    const int per_split = ksteps / a.S;
    const dim3 grid(tiles, a.S, zb);
    if constexpr (MODE == SK_RESID) {
      // no cross-workgroup split (the epilogue owns its slice of the residual rows): only N / 16 workgroups, so K goes over 16 or 8 waves
      if (per_split == 160) { hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 1, 2, 16>), grid, dim3(1024), 0, s, a); return 0; }   // (all-loads-first would need > 128 VGPRs at 16 waves)
      if (per_split == 80) { hipLaunchKernelGGL((dec_skinny_fflat<T, MODE, 1, 10, 8>), grid, dim3(512), 0, s, a); return 0; }
      if (per_split == 40) { hipLaunchKernelGGL((dec_skinny_fflat<T, MODE, 1, 5, 8>), grid, dim3(512), 0, s, a); return 0; }
      if (per_split % 16 == 0 && per_split >= 64) { hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 1, 1, 16>), grid, dim3(1024), 0, s, a); return 0; }
      if (per_split % 8 == 0 && per_split >= 16) { hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 1, 1, 8>), grid, dim3(512), 0, s, a); return 0; }
    }
    const int NW = per_split % 4 == 0 ? 4 : per_split % 2 == 0 ? 2 : 1;
    const int nstep = per_split / NW;
#define FF(NS_)                                                                                                       \
    if (NW == 4 && nstep == NS_) { hipLaunchKernelGGL((dec_skinny_fflat<T, MODE, 1, NS_, 4>), grid, dim3(256), 0, s, a); return 0; }
    FF(10) FF(5) FF(8) FF(6) FF(4) FF(3) FF(2) FF(1)
#undef FF
    if (NW == 4) hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 1, 1, 4>), grid, dim3(256), 0, s, a);
    else if (NW == 2) hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 1, 1, 2>), grid, dim3(128), 0, s, a);
    else hipLaunchKernelGGL((dec_skinny_fring<T, MODE, 1, 1, 1>), grid, dim3(64), 0, s, a);
    return 0;
  }
}

template <typename T>
static int skinny_frag_launch_t(const SkinnyArgs& a, int mode, hipStream_t s) {
  switch (mode) {
    case SK_OUT16: return skinny_frag_launch_m<T, SK_OUT16>(a, s);
    case SK_OUTF32: return skinny_frag_launch_m<T, SK_OUTF32>(a, s);
    case SK_PARTIAL: return skinny_frag_launch_m<T, SK_PARTIAL>(a, s);
    case SK_QKV: return skinny_frag_launch_m<T, SK_QKV>(a, s);
    case SK_RESID: return skinny_frag_launch_m<T, SK_RESID>(a, s);
    default: return -1;
  }
}

int dec_launch_skinny(mia_whisper* w, const SkinnyArgs& a, int mode, hipStream_t s) {
  if (!a.A || !a.W || !a.out || a.M <= 0 || a.N <= 0 || a.S <= 0 || a.K % (32 * a.S) != 0) return -1;
  if (mode == SK_QKV && (a.D % 64 != 0 || a.N != 3 * a.D || (a.ldo & 3) || !a.cache_k || !a.cache_v || !a.pos)) return -1;
  if (mode == SK_OUT16 && a.out_frag && a.N % 32 != 0) return -1;
  if (mode == SK_RESID && (a.S != 1 || a.N % 32 != 0 || !a.xres || !a.nw || !a.ss_out)) return -1;
  if (a.c1 && (!a.c2 || !a.ss_in || a.ss_tiles <= 0 || a.ss_dim <= 0)) return -1;
  return w->dtype == MIA_F16 ? skinny_frag_launch_t<F16>(a, mode, s) : skinny_frag_launch_t<BF16>(a, mode, s);
}

int dec_launch_attention(mia_whisper* w, const void* q, const void* kc, const void* vc, void* out, int fixed_keys, int cap_keys,
                         hipStream_t s, float* qk_out, const int32_t* head_slot, int n_slots, int qk_ctx) {
  if (cap_keys > DEC_MAX_KEYS) return -1;
  dim3 grid(w->dims.n_text_head, w->cur_B), block(256);
#define ATT(T_, U_, NTL_)                                                                                                            \
  hipLaunchKernelGGL((dec_attention<T_, U_, NTL_>), grid, block, 0, s, (const uint16_t*)q, (const uint16_t*)kc, (const uint16_t*)vc,   \
                     (uint16_t*)out, w->clip.pos, fixed_keys, cap_keys, w->dims.n_text_head, 0.125f, qk_out, head_slot, n_slots, qk_ctx)
  // cross attention (fixed_keys = the 1500 encoder positions): bandwidth-bound streaming of 2 x 192 KB per (clip, head), non-temporal;
  // self attention: at most n_text_ctx cached keys, latency-bound.  (U = 8 was measured: 10 us per step SLOWER than U = 4.)
  if (w->dtype == MIA_F16) { if (fixed_keys > 0) ATT(F16, 4, true); else ATT(F16, 4, false); }
  else { if (fixed_keys > 0) ATT(BF16, 4, true); else ATT(BF16, 4, false); }
#undef ATT
  return 0;
}

// true when the greedy head runs as partial + final kernels; the final kernel then also embeds the next position, so the step graph
// carries no dec_embed_ln of its own (the caller launches it once before the first step)
bool dec_head_is_split(const DecodeParams& p) {
  return p.greedy && !p.head_single && p.V <= 256 * HEAD_NPT2 * HEAD_SPLIT;
}

// test hook (mia_whisper_trace_logits): the raw logits the head is about to read, for the traced clips, filed under the position of
// the token this step consumed.  Runs inside the captured step graph, so what is traced is what the graph computed.
__global__ __launch_bounds__(256) void dec_trace_logits(const float* __restrict__ logits, const int32_t* __restrict__ pos, const int32_t* __restrict__ clips,
                                                        float* __restrict__ trace, int V, int n_ctx) {
  const int slot = blockIdx.y, b = clips[slot];
  const int p = pos[b];
  if (p < 0 || p >= n_ctx) return;
  const float* src = logits + (int64_t)b * V;
  float* dst = trace + ((int64_t)slot * n_ctx + p) * V;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) dst[i] = src[i];
}

int dec_launch_trace(mia_whisper* w, hipStream_t s) {
  if (!w->trace || w->trace_n <= 0) return -1;
  hipLaunchKernelGGL(dec_trace_logits, dim3(32, w->trace_n), dim3(256), 0, s, w->logits, w->clip.pos, w->trace_clips, w->trace, w->dims.n_vocab, w->dims.n_text_ctx);
  return 0;
}

int dec_launch_head(mia_whisper* w, int32_t* last_ts, const DecodeParams& p, hipStream_t s) {
  if (p.V > 1024 * HEAD_NPT) return -1;
  HeadBufs hb{w->logits, w->tokens, w->n_gen, w->finished, last_ts, w->sum_logprob, w->n_logprob, w->no_speech, w->suppress_bits, w->uniforms, w->clip};
  if (dec_head_is_split(p)) {
    // argmax path: the vocabulary is reduced by HEAD_SPLIT workgroups per clip, then merged (w->partial is free at this point of the step)
    HeadPart* parts = reinterpret_cast<HeadPart*>(w->partial);
    hipLaunchKernelGGL(dec_head_partial, dim3(HEAD_SPLIT, w->cur_B), dim3(256), 0, s, hb, p, parts);
    const LNW& ln0 = w->dec[0].attn_ln;
    NextEmbed ne{(const uint16_t*)w->tok_emb, w->dec_pos, ln0.g, ln0.b, w->dx, (uint16_t*)w->dh, w->dims.n_text_state};
    if (w->dtype == MIA_F16) hipLaunchKernelGGL(dec_head_final<F16>, dim3(w->cur_B), dim3(256), 0, s, hb, p, (const HeadPart*)parts, ne);
    else hipLaunchKernelGGL(dec_head_final<BF16>, dim3(w->cur_B), dim3(256), 0, s, hb, p, (const HeadPart*)parts, ne);
    return 0;
  }
  hipLaunchKernelGGL(dec_head, dim3(w->cur_B), dim3(1024), 0, s, hb, p);
  return 0;
}

int dec_launch_finalize(mia_whisper* w, int32_t* out_n, const DecodeParams& p, hipStream_t s) {
  hipLaunchKernelGGL(dec_finalize, dim3(w->cur_B), dim3(256), 0, s, w->tokens, w->n_gen, w->sum_logprob, w->n_logprob, w->out_tokens, out_n, w->out_avg, w->clip.n_init, p);
  return 0;
}
