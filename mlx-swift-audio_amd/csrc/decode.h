// decode.h -- internal interface between whisper_decode.hip (host loop) and decode_kernels.hip.
#pragma once
#include "whisper.h"

enum { SK_OUT16 = 0, SK_OUTF32 = 1, SK_PARTIAL = 2, SK_QKV = 3, SK_SWIGLU = 4, SK_RESID = 5 };

// ---- MFMA-fragment order (the Whisper decode step's GEMM operands) -------------------------------------------------------------
// v_mfma_f32_16x16x32 takes, per lane (r = lane & 15, c = lane >> 4), the 8 consecutive K-values 8c..8c+7 of row r.  Stored row-major,
// one wave instruction gathers 16 rows x 64 B (16 half cache lines whose 2560-byte stride maps them onto a few L2 channels); stored in
// fragment order it is ONE contiguous 1 KB.  tools/micro/skinny_probe.hip measures the difference on the decoder's fc1 shape
// (N 5120, K 1280, 32 rows): 11.9 us row-major -> 10.0 us with the weights in fragment order -> 8.3 us with the activations too.
//   activations [32 z][K]:  element (m, k) -> ((((m >> 5) * (K >> 5) + (k >> 5)) * 2 + ((m >> 4) & 1)) * 64 + ((k >> 3) & 3) * 16 + (m & 15)) * 8 + (k & 7)
//   weights     [N][K]:     element (n, k) -> (((n >> 4) * (K >> 5) + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7), rows padded to 16
__host__ __device__ inline int64_t afrag_index(int m, int k, int K) {
  return ((((int64_t)(m >> 5) * (K >> 5) + (k >> 5)) * 2 + ((m >> 4) & 1)) * 64 + ((k >> 3) & 3) * 16 + (m & 15)) * 8 + (k & 7);
}
__host__ __device__ inline int64_t wfrag_index(int n, int k, int K) {
  return (((int64_t)(n >> 4) * (K >> 5) + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7);
}

struct SkinnyArgs {
  const uint16_t* A; int64_t lda;        // [M][K]
  const uint16_t* W;                     // [N][K]
  const float* bias;
  void* out; int64_t ldo;                // OUT16 / OUTF32: [M][ldo];  PARTIAL: [S][M][N];  QKV: q [M][D]
  uint16_t* cache_k; uint16_t* cache_v;  // QKV: [B][H][n_ctx][64] of this layer
  const int32_t* pos;                    // QKV: per-row (clip) cache position
  int M, N, K, S, act, D, H, n_ctx;
  int out_frag = 0;                      // fragment-order kernels, OUT16: write `out` in activation fragment order (row length N)
  int w_frag = 0;                        // row-major-activation kernels (the LM step): W is in weight fragment order
  int w_keep = 0;                        // fragment-order kernels (the Whisper step): load W with the default cache policy instead of
                                         // non-temporal -- several decode loops read one weight copy concurrently (mia_whisper_set_weight_sharing)
  // ---- RMSNorm carried across the GEMM chain of the LM step (no separate reduce + norm launch):
  // SK_RESID (needs S == 1): xres[m][n] += acc (+ bias) in place (the fp32 residual stream, row stride N); out[m][n] (16-bit, row stride
  //   ldo) = (x_new * nw[n]) rounded -- the NEXT block's activation, not yet divided by its rms; ss_out[tile][m] = sum of x_new^2 over the
  //   tile's 16 columns (tile = n / 16), one value per (tile, row), written by exactly one lane: a fixed-order partial sum.
  // consumers of such an activation (any mode) pass ss_in / ss_tiles / ss_dim / eps: the accumulator of row m is multiplied by
  //   rstd[m] = rsqrt(sum_t ss_in[t][m] / ss_dim + eps) before bias and activation -- linear, so it commutes with the contraction.
  float* xres = nullptr; const float* nw = nullptr; float* ss_out = nullptr;
  const float* ss_in = nullptr; int ss_tiles = 0; int ss_dim = 0; float eps = 0.f;
  // LayerNorm form (the Whisper step, fragment-order kernels): ss_* hold PAIRS (sum x, sum x^2) per (tile, row); a consumer passes the
  // Linear's folded constants c1[n] = sum_k W[n][k] gamma[k], c2[n] = sum_k W[n][k] beta[k] and its epilogue computes
  //   LN(x) W^T = rstd (acc - mean c1) + c2      with acc = W (x * gamma), the activation the SK_RESID producer stored
  const float* c1 = nullptr; const float* c2 = nullptr;
};

int dec_launch_embed_ln(mia_whisper* w, const LNW& ln, hipStream_t s);
int dec_launch_reduce_ln(mia_whisper* w, int S, const float* bias, const LNW& ln, hipStream_t s);
// Whisper step: a.A in activation fragment order, a.W in weight fragment order (LinearW::wf)
int dec_launch_skinny(mia_whisper* w, const SkinnyArgs& a, int mode, hipStream_t s);
// c1[n] = sum_k W[n][k] gamma[k], c2[n] = sum_k W[n][k] beta[k] for a row-major 16-bit [N][K] matrix (fp32 sums in k order)
int dec_launch_lnfold(const void* w16, int N, int K, const float* gamma, const float* beta, float* c1, float* c2, int dtype, hipStream_t s, const float* bias = nullptr);
// row-major [N][K] 16-bit -> weight fragment order (dst holds ceil(N/16)*16*K elements; rows past N are zero)
int dec_launch_repack_wfrag(const void* src, void* dst, int N, int K, hipStream_t s);
// model-independent form (dtype = MIA_BF16 | MIA_F16); SK_SWIGLU: W rows interleaved gate/up, out[m][n/2] = silu(g)*u (16-bit)
int skinny_gemm_launch(const SkinnyArgs& a, int mode, int dtype, hipStream_t s);
// MLX-affine 4- / 8-bit weights in fragment order (decode_kernels.hip: skinny_gemm_qi); modes SK_OUT16 / SK_OUTF32 / SK_PARTIAL / SK_SWIGLU
int skinny_gemm_q_launch(const SkinnyArgs& a, const uint32_t* wfrag, const float* stfrag, int bits, int mode, int dtype, hipStream_t s);
// qk_out (optional): pre-softmax scores of the heads with head_slot[h] >= 0 -> qk_out[b][slot][pos[b]][key] (word-timestamp alignment)
int dec_launch_attention(mia_whisper* w, const void* q, const void* kc, const void* vc, void* out, int fixed_keys, int cap_keys,
                         hipStream_t s, float* qk_out = nullptr, const int32_t* head_slot = nullptr, int n_slots = 0, int qk_ctx = 0);
bool dec_head_is_split(const DecodeParams& p);
// test hook: logits of the traced clips -> w->trace[slot][pos[clip]][0:V]
int dec_launch_trace(mia_whisper* w, hipStream_t s);
int dec_launch_head(mia_whisper* w, int32_t* last_ts, const DecodeParams& p, hipStream_t s);
int dec_launch_finalize(mia_whisper* w, int32_t* out_n, const DecodeParams& p, hipStream_t s);
