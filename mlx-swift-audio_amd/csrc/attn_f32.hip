// attn_f32.hip -- fp32 flash attention (head dim 64) on the exact-fp32 matrix cores, gfx950.
//
// Used by the CosyVoice2 flow: the CFM estimator's DiffusersAttention (softmax(q k^T / 8) v,
// Codec/S3Gen/Matcha/MatchaTransformer.swift:36-77) and the conformer encoder's RelPositionMultiHeadedAttention
// (Codec/S3Gen/Transformer/Attention.swift:143-195).  The latter's score is ((q + u) k_j + (q + v) p_j) / 8 with p = linear_pos(pe):
// the reference builds the encoder with the one-sided RelPositionalEncoding, so matrix_bd needs no rel_shift and the positional
// term is one more 64-wide contraction against per-position keys: the POS variant runs the QK product over the concatenated
// 128-wide operands [q + u | q + v] . [k | p].
//
// One 256-thread workgroup = 32 queries of one (batch, head), its 4 waves each walking a quarter of the key tiles (split-KV, merged
// through LDS in wave order).  Work units are small on purpose: with 64- or 128-query workgroups T = 1050 gave 272 / 144 workgroups for 256
// CUs and the kernel time was set by the CUs holding two of them (measured 90 us at T = 1050 vs 30 us at T = 525).  S^T = K Q^T on v_mfma_f32_32x32x2_f32, so
// a lane owns ONE query column and 16 keys of it: the online softmax is in-lane plus one cross-half shuffle, and the S^T
// accumulator registers are, in order, the B operand of O^T += V^T P^T (the A operand reads V rows in the matching key order).
// K / P / V tiles of 32 keys are staged HBM -> registers -> LDS with row strides (66 | 130, 72 floats) that make every
// fragment read bank-conflict free.
#include "mia_device.h"
#include "ops.h"

namespace {

template <bool POS>
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF32Args a) {
  constexpr int DQK = POS ? 128 : 64;
  constexpr int SK = DQK + 2;       // K row stride (floats): 2*key + half distinct mod 64
  constexpr int SV = 72;            // V row stride: keys k and k+4 land 32 banks apart
  // four KV tiles per iteration, one per wave
  constexpr int NKV = 4;
  __shared__ __attribute__((aligned(16))) float lds[NKV * 32 * SK + NKV * 32 * SV];   // K tiles | V tiles; reused for the Q tile and the merge
  float (*Ks)[32 * SK] = reinterpret_cast<float (*)[32 * SK]>(lds);
  float (*Vs)[32 * SV] = reinterpret_cast<float (*)[32 * SV]>(lds + NKV * 32 * SK);
  static_assert((NKV - 1) * 64 * 34 <= NKV * 32 * SK + NKV * 32 * SV, "merge buffer must fit in the tile storage");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kvh = wave;
  const int lq = lane & 31, lh = lane >> 5;
  const int nqb = (a.T + 31) / 32;
  const int qb = blockIdx.x % nqb, hb = blockIdx.x / nqb;
  const int h = hb % a.H, b = hb / a.H;
  const int q0 = qb * 32;
  const float* Q = a.q + (int64_t)b * a.T * a.ldq + h * 64;
  const float* K = a.k + (int64_t)b * a.T * a.ldk + h * 64;
  const float* V = a.v + (int64_t)b * a.T * a.ldv + h * 64;
  const float* P = POS ? a.p + h * 64 : nullptr;

  // ---- Q^T fragments: lane holds (q + bias)[2 i + lh] * scale * log2(e), i = 0..31 (and the +v copy for POS).  The 64 x 64 query tile is
  // first read coalesced (float4 per thread) into LDS -- per-lane strided 8-byte reads of 32 different rows serialised in the TA.
  float qf[DQK / 2];
  {
    float* Qs = lds;                                   // [32][66], dead before the first K / V tile is written
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (tid >> 4) + 16 * j, c4 = (tid & 15) * 4;
      int q = q0 + row; q = q < a.T ? q : a.T - 1;
      const f32x4 v = *reinterpret_cast<const f32x4*>(Q + (int64_t)q * a.ldq + c4);
      float* d = Qs + row * 66 + c4;
      *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v[2], v[3]);
    }
    __syncthreads();
    const float* qp = Qs + lq * 66 + lh;
    const float sc = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float x = qp[2 * i];
      if (POS) {
        qf[i] = (x + a.bias_u[h * 64 + 2 * i + lh]) * sc;
        qf[32 + i] = (x + a.bias_v[h * 64 + 2 * i + lh]) * sc;
      } else {
        qf[i] = x * sc;
      }
    }
  }

  // ---- staging: 4 tiles x 32 keys x 64 floats = 2048 float4 per source, 8 per thread: thread -> (tile = tid >> 6, rows (tid & 63) >> 4 + 4 j, 16-B column)
  constexpr int NST = 8;
  const int s_tile = tid >> 6, s_row = (tid & 63) >> 4, s_c4 = (tid & 15) * 4;
  f32x4 rk[NST], rp[NST], rv[NST];   // native vectors: HIP's float4 struct copies were lowered to memcpy through a scratch array
  auto load_regs = [&](int key0) {
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      int key = key0 + 32 * s_tile + s_row + 4 * j; key = key < a.T ? key : a.T - 1;      // (rows past the walk's end are clamped reads, masked later)
      rk[j] = *reinterpret_cast<const f32x4*>(K + (int64_t)key * a.ldk + s_c4);
      rv[j] = *reinterpret_cast<const f32x4*>(V + (int64_t)key * a.ldv + s_c4);
      if (POS) rp[j] = *reinterpret_cast<const f32x4*>(P + (int64_t)key * a.ldp + s_c4);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int row = s_row + 4 * j;
      float* kd = Ks[s_tile] + row * SK + s_c4;              // rows are 8-byte aligned only
      *reinterpret_cast<float2*>(kd) = make_float2(rk[j][0], rk[j][1]);
      *reinterpret_cast<float2*>(kd + 2) = make_float2(rk[j][2], rk[j][3]);
      if (POS) {
        *reinterpret_cast<float2*>(kd + 64) = make_float2(rp[j][0], rp[j][1]);
        *reinterpret_cast<float2*>(kd + 66) = make_float2(rp[j][2], rp[j][3]);
      }
      *reinterpret_cast<f32x4*>(Vs[s_tile] + row * SV + s_c4) = rv[j];
    }
  };

  f32x16 acc_o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // streaming mask: this lane's query sees keys below klim; the workgroup's key walk stops at its last query's limit
  // (padded batch: keys at or beyond the sequence's own length Tb are masked like a chunk limit; the rows exist -- finite padding --
  // so their probability is an exact 0 and the walk's length, tiles and order are those of the sequence's own B = 1 call)
  const int Tb = a.seq_len ? min(a.T, max(1, a.seq_len[b])) : a.T;
  const int qmine = q0 + lq < a.T ? q0 + lq : a.T - 1;
  const int klim = a.chunk > 0 ? min(Tb, (qmine / a.chunk + 1) * a.chunk) : Tb;
  const int qlast = q0 + 31 < a.T ? q0 + 31 : a.T - 1;
  const int Tk = a.chunk > 0 ? min(Tb, (qlast / a.chunk + 1) * a.chunk) : Tb;       // workgroup-uniform
  const int npairs = (Tk + 32 * NKV - 1) / (32 * NKV);
  load_regs(0);
  for (int kp = 0; kp < npairs; ++kp) {
    const int key0 = kp * 32 * NKV + 32 * kvh;    // this wave's tile
    __syncthreads();                              // previous tiles fully consumed
    write_lds();
    __syncthreads();
    if (kp + 1 < npairs) load_regs((kp + 1) * 32 * NKV);
    if (key0 < Tk) {                              // (the odd tail tile may be absent: wave-uniform)

    // ---- S^T = K Q^T
    f32x16 acc_s;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_s[r] = 0.f;
    // LDS operand reads are batched 8 MFMA steps ahead (sched_barrier keeps the compiler from sinking each read next to its MFMA,
    // which exposed the full LDS latency on every one of the dependent MFMAs)
    const float* kp_ = Ks[kvh] + lq * SK + lh;
    {
      constexpr int NB = DQK / 16;
      float kf[2][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kf[0][u] = kp_[2 * u];
#pragma unroll
      for (int bq = 0; bq < NB; ++bq) {
        const int cur = bq & 1;
        if (bq + 1 < NB) {
#pragma unroll
          for (int u = 0; u < 8; ++u) kf[cur ^ 1][u] = kp_[2 * (8 * (bq + 1) + u)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc_s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cur][u], qf[8 * bq + u], acc_s, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- online softmax (base 2): lane = one query, registers = keys (r&3) + 8 (r>>2) + 4 lh.  The probabilities go to their own
    // scalar array: writing elements back into the MFMA result vector made the compiler round-trip it through scratch memory.
    float pr[16];
    float mloc = -INFINITY;
    const bool tail = key0 + 32 > klim;           // per lane: keys at or beyond this query's limit are masked (klim == T without chunks)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sv = acc_s[r];
      if (tail && key0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= klim) sv = -INFINITY;
      pr[r] = sv;
      mloc = fmaxf(mloc, sv);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    // a tile can be entirely beyond ONE lane's limit (chunk mask): keep that lane's state at "nothing seen" without producing NaNs
    const float m_ref = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_ref);
    m_run = m_new;
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = __builtin_amdgcn_exp2f(pr[r] - m_ref);
      lsum += pr[r];
    }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;

    // ---- O^T += V^T P^T: step j contracts key (j&3) + 8 (j>>2) + 4 lh
    {
      const float* vbase = Vs[kvh] + 4 * lh * SV + lq;
      float vf[2][4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) { vf[0][u][0] = vbase[u * SV]; vf[0][u][1] = vbase[u * SV + 32]; }
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {       // batch bq = steps j = 4 bq .. 4 bq + 3 = key rows 8 bq + (0..3) + 4 lh
        const int cur = bq & 1;
        if (bq + 1 < 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) { vf[cur ^ 1][u][0] = vbase[(8 * (bq + 1) + u) * SV]; vf[cur ^ 1][u][1] = vbase[(8 * (bq + 1) + u) * SV + 32]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc_o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[cur][u][0], pr[4 * bq + u], acc_o[0], 0, 0, 0);
          acc_o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[cur][u][1], pr[4 * bq + u], acc_o[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
  }

  // ---- merge the four KV quarters in wave order (fixed order: deterministic), normalise, store
  l_run += __shfl_xor(l_run, 32, 64);             // row sum over this wave's keys, in both lane halves
  __syncthreads();                                // K / V tiles are dead: reuse the storage as the hand-over buffer
  if (kvh > 0) {
    float* mb = lds + ((kvh - 1) * 64 + lane) * 34;   // [3 waves][64 lanes][m, l, 32 x o]
    mb[0] = m_run; mb[1] = l_run;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) mb[2 + db * 16 + r] = acc_o[db][r];
  }
  __syncthreads();
  if (kvh > 0) return;
  {
    float m = m_run;
#pragma unroll
    for (int w2 = 0; w2 < NKV - 1; ++w2) m = fmaxf(m, lds[(w2 * 64 + lane) * 34]);
    const float a0 = m_run == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_run - m);
    float den = l_run * a0;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] *= a0;
#pragma unroll
    for (int w2 = 0; w2 < NKV - 1; ++w2) {
      const float* mb = lds + (w2 * 64 + lane) * 34;
      const float mw = mb[0];
      const float aw = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - m);
      den += mb[1] * aw;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] += mb[2 + db * 16 + r] * aw;
    }
    const float inv = 1.0f / den;
    const int q = q0 + lq;
    if (q < a.T) {
      float* op = a.out + ((int64_t)b * a.T + q) * a.ldo + h * 64 + 4 * lh;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(op + db * 32 + 8 * g) =
              make_float4(acc_o[db][4 * g] * inv, acc_o[db][4 * g + 1] * inv, acc_o[db][4 * g + 2] * inv, acc_o[db][4 * g + 3] * inv);
    }
  }
}

}  // namespace

const char* mia_attn_f32_check(const AttnF32Args& a) {
  if (a.B <= 0 || a.T <= 0 || a.H <= 0) return "attn_f32: B, T, H must be > 0";
  if (!a.q || !a.k || !a.v || !a.out) return "attn_f32: null operand";
  if (a.ldq % 4 || a.ldk % 4 || a.ldv % 4 || a.ldo % 4) return "attn_f32: row strides must be multiples of 4 floats";
  if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.out) & 15) return "attn_f32: operands must be 16-byte aligned";
  if (a.p && (!a.bias_u || !a.bias_v || a.ldp % 4 || ((uintptr_t)a.p & 15))) return "attn_f32: positional keys need both biases and aligned rows";
  return nullptr;
}

int mia_attn_f32_launch(const AttnF32Args& a, hipStream_t s) {
  dim3 grid(((a.T + 31) / 32) * a.H * a.B), block(256);
  if (a.p) hipLaunchKernelGGL(attn_f32_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(attn_f32_kernel<false>, grid, block, 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
