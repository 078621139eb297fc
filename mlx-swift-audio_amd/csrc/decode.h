// decode.h -- internal interface between whisper_decode.hip (host loop) and decode_kernels.hip.
#pragma once
#include "whisper.h"

enum { SK_OUT16 = 0, SK_OUTF32 = 1, SK_PARTIAL = 2, SK_QKV = 3, SK_SWIGLU = 4 };

struct SkinnyArgs {
  const uint16_t* A; int64_t lda;        // [M][K]
  const uint16_t* W;                     // [N][K]
  const float* bias;
  void* out; int64_t ldo;                // OUT16 / OUTF32: [M][ldo];  PARTIAL: [S][M][N];  QKV: q [M][D]
  uint16_t* cache_k; uint16_t* cache_v;  // QKV: [B][H][n_ctx][64] of this layer
  const int32_t* pos;                    // QKV: per-row (clip) cache position
  int M, N, K, S, act, D, H, n_ctx;
};

int dec_launch_embed_ln(mia_whisper* w, const LNW& ln, hipStream_t s);
int dec_launch_reduce_ln(mia_whisper* w, int S, const float* bias, const LNW& ln, hipStream_t s);
int dec_launch_skinny(mia_whisper* w, const SkinnyArgs& a, int mode, hipStream_t s);
// model-independent form (dtype = MIA_BF16 | MIA_F16); SK_SWIGLU: W rows interleaved gate/up, out[m][n/2] = silu(g)*u (16-bit)
int skinny_gemm_launch(const SkinnyArgs& a, int mode, int dtype, hipStream_t s);
// MLX-affine 4-bit weights in fragment order (decode_kernels.hip: skinny_gemm_q4); modes SK_OUT16 / SK_OUTF32 / SK_PARTIAL / SK_SWIGLU
int skinny_gemm_q4_launch(const SkinnyArgs& a, const uint32_t* wfrag, const uint16_t* sbfrag, int scale_dtype, int mode, int dtype, hipStream_t s);
// qk_out (optional): pre-softmax scores of the heads with head_slot[h] >= 0 -> qk_out[b][slot][pos[b]][key] (word-timestamp alignment)
int dec_launch_attention(mia_whisper* w, const void* q, const void* kc, const void* vc, void* out, int fixed_keys, int cap_keys,
                         hipStream_t s, float* qk_out = nullptr, const int32_t* head_slot = nullptr, int n_slots = 0, int qk_ctx = 0);
bool dec_head_is_split(const DecodeParams& p);
int dec_launch_head(mia_whisper* w, int32_t* last_ts, const DecodeParams& p, hipStream_t s);
int dec_launch_finalize(mia_whisper* w, int32_t* out_n, const DecodeParams& p, hipStream_t s);
