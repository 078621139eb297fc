// gemm.h -- internal interface of the MFMA GEMM (gemm.hip) and the skinny decode GEMM (gemm_skinny.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mia.h"

enum { MIA_ACT_NONE = 0, MIA_ACT_GELU = 1 };
enum { MIA_EPI_STD = 0, MIA_EPI_QKV_VT = 1, MIA_EPI_HEADMAJOR = 2 };

// C[bz][m][n] = act(sum_k A[bz][m][k] * W[n][k] + bias[n]) + R[bz][m][n]
struct GemmArgs {
  const void* A = nullptr;   // 16-bit [batch][M][lda]   (rows may overlap: lda < K is legal, used by the convs)
  const void* W = nullptr;   // 16-bit [N][K], K contiguous
  void* C = nullptr;         // 16-bit or fp32 [batch][M][ldc]
  void* C2 = nullptr;        // QKV_VT: transposed V destination [B][H][64][Tpad]
  const float* bias = nullptr;
  const float* R = nullptr;  // fp32 residual / positional table [batch][M][ldr]
  int64_t lda = 0, strideA = 0;
  int64_t ldc = 0, strideC = 0;
  int64_t ldr = 0, strideR = 0;
  int M = 0, N = 0, K = 0, batch = 1;
  int act = MIA_ACT_NONE;
  int out_f32 = 0;
  int epi = MIA_EPI_STD;
  int T = 0, H = 0, Tpad = 0;   // special epilogues: m = b*T + t, n = h*64 + d
  int variant = 3;              // 0: 128^2 register-staged, 1: 128^2 LDS-DMA staged, 2: 256^2 two-buffer LDS-DMA, 4: 256^2 8-phase LDS-DMA ring, 3: auto (4 when it fills the chip, else 1)
};

// returns nullptr when the arguments satisfy the kernel's shape/alignment assumptions, else a message
const char* mia_gemm_check(const GemmArgs& g);
// dtype: MIA_BF16 or MIA_F16 (type of A, W and of a 16-bit C)
int mia_gemm_launch(const GemmArgs& g, int dtype, hipStream_t s);
