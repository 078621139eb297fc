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
