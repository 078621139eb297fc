// attention.hip -- fused (flash-style) multi-head attention for the Whisper encoder, head dim 64, gfx950.
//
// Replaces WhisperMultiHeadAttention.qkvAttention (STT/Whisper/Layers/MultiHeadAttention.swift:85-135) for the
// encoder's full (unmasked) self-attention: softmax((q s)(k s)^T) v with s = d_h^-1/4 on both operands, fp32
// softmax ("precise").  The raw qk tensor the Swift returns is only used by word timestamps, not on this path.
//
// Structure (one 256-thread workgroup = 128 queries of one (clip, head); one wave = 32 queries):
//   * S^T = K Q^T on v_mfma_f32_32x32x16 ("swapped" product): a lane then owns ONE query column and 32 keys of
//     it in registers, so the row max / row sum of the online softmax are in-lane plus one cross-half shuffle;
//   * the S^T accumulator is consumed directly as the B operand of O^T = V^T P^T (no LDS round trip): K rows are
//     fed with bits 2<->3 of the row index swapped so the accumulator's register order equals natural key order;
//   * K tiles [64 keys][64] and V^T tiles [64 d][64 keys] are staged HBM -> registers -> LDS (issue early, write
//     after the barrier, double buffered), 128-B rows with the 16-B chunk index XOR-swizzled by (row>>1)&7 so
//     all fragment reads (ds_read_b128) are bank-conflict free;
//   * V arrives pre-transposed ([B][H][64][Tpad], zero padded) from the QKV GEMM epilogue (gemm.hip QKV_VT).
#include <type_traits>

#include "mia_device.h"
#include "ops.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int TILE_BYTES = 64 * 128;  // 8 KB

__device__ __forceinline__ int swap23(int r) {  // swap bits 2 and 3
  return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void enc_attention_kernel(const uint16_t* __restrict__ qk, int64_t ld_qk,
                                                               const uint16_t* __restrict__ vt, uint16_t* __restrict__ out,
                                                               int64_t ld_out, int T_len, int Tpad, int H, float scale_log2) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // [2 buffers][K | V^T]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware mapping: the q-blocks of one (clip, head) re-read the same K / V^T; consecutive block ids are dealt round-robin to
  // the 8 XCDs, so give every XCD a contiguous run of the linear index and keep the q-blocks of a (clip, head) adjacent in it.
  const int nqb = (T_len + 127) / 128;                   // launch is 1-D: gridDim.x = nqb * H * B
  int lin = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = lin % 8;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + lin / 8;
  }
  const int qb = lin % nqb, hb = lin / nqb;
  const int h = hb % H, b = hb / H;
  const int D = H * 64;
  const int q0 = qb * 128 + wave * 32;
  const int lq = lane & 31, lh = lane >> 5;

  // ---- Q^T fragments (B operand): lane holds Q[q0+lq][16*ks + 8*lh + 0..7]
  s16x8 qf[4];
  {
    int q = q0 + lq; q = q < T_len ? q : T_len - 1;
    const uint16_t* qp = qk + ((int64_t)b * T_len + q) * ld_qk + h * 64 + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const s16x8*>(qp + 16 * ks);
  }

  // ---- staging: 2 x 16 B of K and 2 x 16 B of V^T per thread per tile
  const int s_row = tid >> 3, s_chk = tid & 7;
  const uint16_t* kbase = qk + (int64_t)b * T_len * ld_qk + D + h * 64 + s_chk * 8;
  const uint16_t* vbase = vt + ((int64_t)b * H + h) * 64 * Tpad + s_chk * 8;
  int s_dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = s_row + 32 * i;
    s_dst[i] = row * 128 + ((s_chk ^ ((row >> 1) & 7)) << 4);
  }
  u32x4 rk[2], rv[2];
  auto load_regs = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = s_row + 32 * i;
      int key = key0 + row; key = key < T_len ? key : T_len - 1;   // masked below; any finite value will do
      rk[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)key * ld_qk);
      rv[i] = *reinterpret_cast<const u32x4*>(vbase + (int64_t)row * Tpad + key0);
    }
  };
  auto write_lds = [&](int buf) {
    char* base = lds + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(base + s_dst[i]) = rk[i];
      *reinterpret_cast<u32x4*>(base + TILE_BYTES + s_dst[i]) = rv[i];
    }
  };

  f32x16 acc_o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int krow = swap23(lq);   // K row fed to MFMA row lq
  const int ntiles = (T_len + KV_TILE - 1) / KV_TILE;
  // One K/V tile.  TAIL (the last, partial tile only) masks the keys at or beyond T_len: in the full tiles the masking selects are
  // not compiled at all (as a runtime `if (tail)` hipcc if-converted them into ~170 of the tile's ~480 instructions, every tile).
  // The softmax works on the RAW scores: max over s, then p = exp2(fma(s, c, -m c)) with c = scale * log2(e) > 0 -- one fma + one
  // exp per score instead of mul, sub, exp; the reference point m is kept raw.
  auto tile = [&](int kt, int cur, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int key0 = kt * KV_TILE;
    if (kt + 1 < ntiles) load_regs(key0 + KV_TILE);
    const char* sk = lds + cur * 2 * TILE_BYTES;
    const char* sv = sk + TILE_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks
    f32x16 acc_s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_s[kb][r] = 0.f;
      const int row = kb * 32 + krow;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const s16x8 kf = *reinterpret_cast<const s16x8*>(sk + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
        acc_s[kb] = T::mfma32(kf, qf[ks], acc_s[kb]);
      }
    }
    // ---- online softmax (base-2), lane = one query, registers = keys
    float mloc = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (TAIL) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key0 + kb * 32 + swap23(i) >= T_len) acc_s[kb][r] = -INFINITY;
        }
        mloc = fmaxf(mloc, acc_s[kb][r]);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // Bounded-growth lazy rescaling: the reference point m_run of a query moves only when the tile's maximum exceeds it by more than
    // 2^LAZY_LOG2 (in the exponent's units), so p = exp2(c (s - m_run)) <= 2^LAZY_LOG2 and the O / l rescale -- 32 multiplies + one exp
    // per tile, ~12 % of the loop's vector issue -- runs in the first tile and then almost never (softmax is invariant to the reference
    // point; the 16-bit P keeps its relative precision at any scale, O and l accumulate in fp32).  The test is wave-uniform.
    constexpr float LAZY_LOG2 = 8.0f;
    const bool grow = (mloc - m_run) * scale_log2 > LAZY_LOG2;   // first tile: m_run = -inf; every tile holds >= 1 valid key
    if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
      const float m_new = grow ? mloc : m_run;
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);       // 1 for the lanes that stay
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[db][r] *= alpha;
    }
    const float neg_mc = -m_run * scale_log2;
    // (pairing the scores through v_pk_fma_f32 / v_pk_add_f32 was measured: 231 instead of 256 instructions per tile, 16.5-16.6 ms
    // per pass against 16.3-16.4 in alternating runs on one box -- the packed fp32 ops are not double rate here)
    float lsum = 0.f;
    s16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(fmaf(acc_s[kb][r], scale_log2, neg_mc));
        lsum += p[r];
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2<T>(p[8 * s + 2 * j], p[8 * s + 2 * j + 1]);
        pf[kb][s] = __builtin_bit_cast(s16x8, w);
      }
    }
    l_run += lsum;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = db * 32 + lq;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int chk = kb * 4 + 2 * s + lh;
          const s16x8 vf = *reinterpret_cast<const s16x8*>(sv + row * 128 + ((chk ^ ((row >> 1) & 7)) << 4));
          acc_o[db] = T::mfma32(vf, pf[kb][s], acc_o[db]);
        }
    }
    if (kt + 1 < ntiles) write_lds(cur ^ 1);
    __syncthreads();
  };
  load_regs(0);
  write_lds(0);
  __syncthreads();
  const int nfull = T_len / KV_TILE;
  int cur = 0;
  for (int kt = 0; kt < nfull; ++kt) { tile(kt, cur, std::false_type{}); cur ^= 1; }
  if (nfull < ntiles) tile(nfull, cur, std::true_type{});

  // ---- normalise and store: lane owns query q0+lq, d = db*32 + (r&3) + 8*(r>>2) + 4*lh
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  const int q = q0 + lq;
  if (q < T_len) {
    uint16_t* op = out + ((int64_t)b * T_len + q) * ld_out + h * 64 + 4 * lh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float o0 = acc_o[db][4 * g + 0] * inv, o1 = acc_o[db][4 * g + 1] * inv;
        const float o2 = acc_o[db][4 * g + 2] * inv, o3 = acc_o[db][4 * g + 3] * inv;
        *reinterpret_cast<u32x2*>(op + db * 32 + 8 * g) = (u32x2){pack2<T>(o0, o1), pack2<T>(o2, o3)};
      }
  }
}

}  // namespace

const char* mia_enc_attention_check(int B, int T, int H, int Tpad, int64_t ld_qk, int64_t ld_out) {
  if (B <= 0 || T <= 0 || H <= 0) return "attention: B, T, H must be > 0";
  if (Tpad % 64 || Tpad < T) return "attention: Tpad must be a multiple of 64 and >= T";
  if (ld_qk % 8 || ld_out % 4) return "attention: row strides must keep 16-byte (qk) / 8-byte (out) alignment";
  return nullptr;
}

int mia_enc_attention_launch(const void* qk, int64_t ld_qk, const void* vt, void* out, int64_t ld_out, int B, int T, int H,
                             int Tpad, int dtype, hipStream_t s) {
  dim3 grid(((T + 127) / 128) * H * B), block(256);
  const float scale_log2 = 0.125f * 1.4426950408889634f;  // (64^-1/4)^2 * log2(e)
  if (dtype == MIA_F16)
    hipLaunchKernelGGL((enc_attention_kernel<F16>), grid, block, 0, s, (const uint16_t*)qk, ld_qk, (const uint16_t*)vt,
                       (uint16_t*)out, ld_out, T, Tpad, H, scale_log2);
  else
    hipLaunchKernelGGL((enc_attention_kernel<BF16>), grid, block, 0, s, (const uint16_t*)qk, ld_qk, (const uint16_t*)vt,
                       (uint16_t*)out, ld_out, T, Tpad, H, scale_log2);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
