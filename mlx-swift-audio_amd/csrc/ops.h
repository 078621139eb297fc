// ops.h -- internal launch interface of the non-GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mia.h"

// ---- norm.hip -------------------------------------------------------------------------------
const char* mia_norm_check(int M, int D, int64_t ldx, int64_t ldy);
int mia_norm_launch(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int M, int D,
                    float eps, bool rms, int out_dtype, hipStream_t s);

// ---- attention.hip (encoder, full attention, d_h = 64) -----------------------------------------
const char* mia_enc_attention_check(int B, int T, int H, int Tpad, int64_t ld_qk, int64_t ld_out);
int mia_enc_attention_launch(const void* qk, int64_t ld_qk, const void* vt, void* out, int64_t ld_out, int B, int T, int H,
                             int Tpad, int dtype, hipStream_t s);

// ---- attn_f32.hip (fp32 flash attention, d_h = 64; optional positional keys) ----------------------
struct AttnF32Args {
  const float* q = nullptr; int64_t ldq = 0;    // [B*T][ldq], head h in columns h*64 .. h*64+63
  const float* k = nullptr; int64_t ldk = 0;
  const float* v = nullptr; int64_t ldv = 0;
  const float* p = nullptr; int64_t ldp = 0;    // optional positional keys [T][ldp] shared by the batch (+ bias_u / bias_v [H][64])
  const float* bias_u = nullptr; const float* bias_v = nullptr;
  float* out = nullptr; int64_t ldo = 0;
  int B = 1, T = 0, H = 0;
  float scale = 0.125f;
  const int32_t* seq_len = nullptr;   // device [B], optional: sequence b holds seq_len[b] <= T valid rows (stacked, padded sequences); keys at or
                                      // beyond it are masked, so a sequence's result equals its own B = 1, T = seq_len[b] call bit for bit
  int chunk = 0;      // > 0: block-causal "streaming" mask, query i sees keys j < (i / chunk + 1) * chunk (subsequentChunkMask,
                      // Codec/S3Gen/Transformer/UpsampleConformerEncoder.swift:124-129); 0 = full attention
};
const char* mia_attn_f32_check(const AttnF32Args& a);
int mia_attn_f32_launch(const AttnF32Args& a, hipStream_t s);
