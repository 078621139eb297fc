// codec.h -- internal interface of the fp32 codec-decoder kernels (codec_kernels.hip) and the layer programs (codec.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mia.h"

// Tap-structured fp32 GEMM: Y[row(m)][n] = epi( sum_{tap,c} pre(X[m + tap*dil - pad][c]) * W[n][tap*Cin + c] + bias[n] )
//   row(m) = m*y_row_mul + y_row_off + z*y_phase_step (rows outside [0,T_out) are dropped); z = grid.z phase, W += z*w_phase_stride
struct ConvGemmArgs {
  const float* X = nullptr; int64_t ldx = 0; int T_in = 0;
  const float* W = nullptr; int64_t w_phase_stride = 0;
  const float* bias = nullptr;
  const float* alpha = nullptr;        // snake prologue on X (per input channel), or null
  const float* ralpha = nullptr;       // snake multiplier (per channel); null -> 1/(alpha + 1e-9) (SNAC / DAC form)
  float lrelu_slope = 0.f;             // > 0: leaky-ReLU prologue on X instead of snake
  float* Y = nullptr; int64_t ldy = 0; int T_out = 0;
  int y_row_mul = 1, y_row_off = 0, y_phase_step = 0;
  const float* R = nullptr; int64_t ldr = 0;   // residual (same row mapping as Y)
  const float* noise = nullptr;        // if set: Y = R + noise[row] * (acc + bias)
  float out_scale = 0.f;               // != 0: Y = out_scale * (act(acc + bias) + R) + R2
  const float* R2 = nullptr;           // second residual, added after the scale (row stride ldr)
  int M = 0, N = 0, Cin = 0, taps = 1, dil = 1, pad = 0;
  int tanh_out = 0;
  int x_row_mul = 1;                   // convolution stride: A(m, tap, c) = X[m*x_row_mul + tap*dil - pad][c]
  int gelu = 0;                        // activation on acc + bias (before the residual): 1 = exact-erf GELU, 2 = ELU, 3 = abs, 4 = SiLU, 5 = leaky-ReLU(0.01), 6 = ReLU
  int64_t ldw = 0;                     // row stride of W (0 = taps*Cin, i.e. dense)
  const int32_t* phase_len = nullptr;  // device [phases], optional (with x_phase_step): sequence z holds phase_len[z] <= T_in valid rows, rows at or
                                       // beyond it read as zero like rows past T_in (padded stacked sequences under a forward-looking window)
  int x_phase_step = 0;                // rows of X skipped per grid.z phase: stacked sequences of T_in rows each (taps never cross
                                       // a sequence; pair with w_phase_stride = 0, y_row_mul = 1, y_phase_step = rows per sequence)
  // Stacked sequences under ANY row mapping (HiFT batches: plain, strided and transposed convolutions): grid.z = n_seq x phases,
  // sequence u = z / phases reads X + u * x_seq_step rows (T_in = rows per sequence, seq_len[u] <= T_in of them valid, the rest read
  // as zero like rows past T_in) and writes rows u * y_seq_step + row(m) of Y (R and R2 likewise); row(m) is tested against T_out
  // per sequence.  n_seq = 1: everything above is unchanged.
  int n_seq = 1;
  int64_t x_seq_step = 0, y_seq_step = 0;
  const int32_t* seq_len = nullptr;    // device [n_seq], optional
};

constexpr int MIA_MAX_LEVELS = 4;
struct EmbedArgs {
  const int32_t* codes[MIA_MAX_LEVELS];   // device, null = level absent
  const float* codebook[MIA_MAX_LEVELS];  // [size][cb_dim]
  const float* weff[MIA_MAX_LEVELS];      // folded weight-normed out_proj [C][cb_dim]
  const float* bias[MIA_MAX_LEVELS];      // [C]
  int stride[MIA_MAX_LEVELS];
  int n_levels, cb_dim;
};

const char* codec_conv_gemm_check(const ConvGemmArgs& g);
// algorithmic bytes of the fp32 conv / elementwise launches issued by the calling host thread since the last reset (profiling aid)
double codec_alg_bytes(bool reset);
int codec_conv_gemm_launch(const ConvGemmArgs& g, int phases, hipStream_t s);
int codec_dwconv_launch(const float* x, float* y, const float* w, const float* bias, const float* a_pre, const float* a_post, int T, int C,
                        int K, int dil, hipStream_t s);
int codec_conv_out1_launch(const float* x, float* out, const float* w, const float* bias, const float* alpha, int T, int C, int K, hipStream_t s);
int codec_embed_launch(const EmbedArgs& a, float* z, int T, int C, hipStream_t s);
int codec_noise1_launch(float* x, const float* w, const float* noise, int T, int C, hipStream_t s);
int codec_conv_in1_launch(const float* x, float* y, const float* w, const float* bias, int64_t T, int C, int K, int pad, hipStream_t s);
int codec_vq_assign_launch(const float* zE, const float* cbn, const float* cbn_sq, const float* cb, const float* weff, const float* bias,
                           float* residual, int32_t* codes, int T, int C, int cs, int cd, hipStream_t s);
