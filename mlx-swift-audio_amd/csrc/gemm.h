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
  // LayerNorm carried across two GEMMs (8-phase kernel only -- ask mia_gemm_ln_ok first): LN(x) W^T = rstd (W (x gamma) - mean c1) + c2
  // with c1[n] = sum_k W[n][k] gamma[k], c2[n] = sum_k W[n][k] beta[k], so the LayerNorm between a residual GEMM and the next Linear
  // needs no pass of its own over the fp32 stream.
  //   producer (out_f32, STD epilogue, N % 64 == 0): besides C = x it stores x * ln_gamma as the next GEMM's 16-bit operand into
  //     ln_out [M][ln_ld] and, per row and 64-column slice, (sum x, sum x^2) into ln_part [M][N/64][2] (mia_ln_finalize_launch turns
  //     them into (mean, rstd) per row);
  //   consumer (16-bit STD output, no residual): C = act(rstd (acc - mean ln_c1[n]) + bias[n]) with (mean, rstd) from ln_stat [M][2];
  //     `bias` then holds c2 + the Linear's own bias.
  const float* ln_gamma = nullptr; void* ln_out = nullptr; int64_t ln_ld = 0; float* ln_part = nullptr;
  const float* ln_stat = nullptr; const float* ln_c1 = nullptr;
};

// true when mia_gemm_launch will run these arguments on the kernel and epilogue form that implement the ln_* fields
bool mia_gemm_ln_ok(const GemmArgs& g);
// (mean, rstd) per row from a producer's ln_part: stat [M][2]
int mia_ln_finalize_launch(const float* part, int n_slices, int D, float eps, float* stat, int M, hipStream_t s);

// returns nullptr when the arguments satisfy the kernel's shape/alignment assumptions, else a message
const char* mia_gemm_check(const GemmArgs& g);
// dtype: MIA_BF16 or MIA_F16 (type of A, W and of a 16-bit C)
int mia_gemm_launch(const GemmArgs& g, int dtype, hipStream_t s);
