// attn_f32.hip -- fp32 flash attention (head dim 64) on the exact-fp32 matrix cores, gfx950.
//
// Used by the CosyVoice2 flow: the CFM estimator's DiffusersAttention (softmax(q k^T / 8) v,
// Codec/S3Gen/Matcha/MatchaTransformer.swift:36-77) and the conformer encoder's RelPositionMultiHeadedAttention
// (Codec/S3Gen/Transformer/Attention.swift:143-195).  The latter's score is ((q + u) k_j + (q + v) p_j) / 8 with p = linear_pos(pe):
// the reference builds the encoder with the one-sided RelPositionalEncoding, so matrix_bd needs no rel_shift and the positional
// term is one more 64-wide contraction against per-position keys: the POS variant runs the QK product over the concatenated
// 128-wide operands [q + u | q + v] . [k | p].
//
// One 256-thread workgroup = 128 queries of one (batch, head); one wave = 32 queries.  S^T = K Q^T on v_mfma_f32_32x32x2_f32, so
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
  __shared__ __attribute__((aligned(16))) float Ks[32 * SK];
  __shared__ __attribute__((aligned(16))) float Vs[32 * SV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int nqb = (a.T + 127) / 128;
  const int qb = blockIdx.x % nqb, hb = blockIdx.x / nqb;
  const int h = hb % a.H, b = hb / a.H;
  const int q0 = qb * 128 + wave * 32;
  const float* Q = a.q + (int64_t)b * a.T * a.ldq + h * 64;
  const float* K = a.k + (int64_t)b * a.T * a.ldk + h * 64;
  const float* V = a.v + (int64_t)b * a.T * a.ldv + h * 64;
  const float* P = POS ? a.p + h * 64 : nullptr;

  // ---- Q^T fragments: lane holds (q + bias)[2 i + lh] * scale * log2(e), i = 0..31 (and the +v copy for POS)
  float qf[DQK / 2];
  {
    int q = q0 + lq; q = q < a.T ? q : a.T - 1;
    const float* qp = Q + (int64_t)q * a.ldq;
    const float sc = a.scale * 1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float2 v2 = *reinterpret_cast<const float2*>(qp + 2 * i);
      const float x = lh ? v2.y : v2.x;
      if (POS) {
        qf[i] = (x + a.bias_u[h * 64 + 2 * i + lh]) * sc;
        qf[32 + i] = (x + a.bias_v[h * 64 + 2 * i + lh]) * sc;
      } else {
        qf[i] = x * sc;
      }
    }
  }

  // ---- staging: tile = 32 keys x 64 floats = 512 float4, 2 per thread per source
  const int s_row = tid >> 4, s_c4 = (tid & 15) * 4;      // rows s_row and s_row + 16
  float4 rk[2], rp[2], rv[2];
  auto load_regs = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = key0 + s_row + 16 * i; key = key < a.T ? key : a.T - 1;
      rk[i] = *reinterpret_cast<const float4*>(K + (int64_t)key * a.ldk + s_c4);
      rv[i] = *reinterpret_cast<const float4*>(V + (int64_t)key * a.ldv + s_c4);
      if (POS) rp[i] = *reinterpret_cast<const float4*>(P + (int64_t)key * a.ldp + s_c4);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = s_row + 16 * i;
      float* kd = Ks + row * SK + s_c4;                     // rows are 8-byte aligned only
      *reinterpret_cast<float2*>(kd) = make_float2(rk[i].x, rk[i].y);
      *reinterpret_cast<float2*>(kd + 2) = make_float2(rk[i].z, rk[i].w);
      if (POS) {
        *reinterpret_cast<float2*>(kd + 64) = make_float2(rp[i].x, rp[i].y);
        *reinterpret_cast<float2*>(kd + 66) = make_float2(rp[i].z, rp[i].w);
      }
      *reinterpret_cast<float4*>(Vs + row * SV + s_c4) = rv[i];
    }
  };

  f32x16 acc_o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (a.T + 31) / 32;
  load_regs(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * 32;
    __syncthreads();                 // previous tile fully consumed
    write_lds();
    __syncthreads();
    if (kt + 1 < ntiles) load_regs(key0 + 32);

    // ---- S^T = K Q^T
    f32x16 acc_s;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_s[r] = 0.f;
    const float* kp = Ks + lq * SK + lh;
#pragma unroll
    for (int i = 0; i < DQK / 2; ++i) acc_s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * i], qf[i], acc_s, 0, 0, 0);

    // ---- online softmax (base 2): lane = one query, registers = keys (r&3) + 8 (r>>2) + 4 lh
    float mloc = -INFINITY;
    const bool tail = key0 + 32 > a.T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = acc_s[r];
      if (tail && key0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= a.T) s = -INFINITY;
      acc_s[r] = s;
      mloc = fmaxf(mloc, s);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(acc_s[r] - m_new);
      acc_s[r] = p;
      lsum += p;
    }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;

    // ---- O^T += V^T P^T: step j contracts key (j&3) + 8 (j>>2) + 4 lh
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float* vp = Vs + ((j & 3) + 8 * (j >> 2) + 4 * lh) * SV + lq;
      acc_o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], acc_s[j], acc_o[0], 0, 0, 0);
      acc_o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], acc_s[j], acc_o[1], 0, 0, 0);
    }
  }

  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
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
  dim3 grid(((a.T + 127) / 128) * a.H * a.B), block(256);
  if (a.p) hipLaunchKernelGGL(attn_f32_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(attn_f32_kernel<false>, grid, block, 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
