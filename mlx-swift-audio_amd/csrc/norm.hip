// norm.hip -- LayerNorm / RMSNorm row kernels: HBM-bound, one wave64 per row,
// 16-byte loads, wavefront shuffle reductions, fp32 statistics, 16-bit (or fp32) output for the next GEMM.
// Replaces MLXNN LayerNorm (eps 1e-5) at ResidualAttentionBlock.swift:65,78,91, AudioEncoder.swift:65,
// TextDecoder.swift:90, and MLXNN RMSNorm at TTS/Orpheus/BuildingBlocks/TransformerBlock.swift:129-139.
#include <cstdlib>
#include "mia_device.h"
#include "ops.h"
#include "gemm.h"

namespace {

// NTX: the rows are streamed once and not needed again soon (the encoder's 246 MB residual stream): non-temporal loads
template <typename T, bool RMS, bool OUT_F32, int NV, bool NTX = false>   // NV float4 per lane cached in registers: D <= 256*NV
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, void* __restrict__ y, int64_t ldy,
                                                   int M, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  const int nv = D >> 2;  // D % 4 == 0 (checked on the host)
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      v[i] = NTX ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + 4 * c)) : *reinterpret_cast<const f32x4*>(xr + 4 * c);
      if (!RMS) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  float mean = 0.f;
  if (!RMS) mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c >= nv) continue;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * gm[j];
    if (beta) {
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += bt[j];
    }
    if (OUT_F32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (int64_t)row * ldy + 4 * c) = o;
    } else {
      *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(y) + (int64_t)row * ldy + 4 * c) =
          (u32x2){pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
    }
  }
}

template <typename T, bool RMS, bool OUT_F32>
void launch_nv(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int M, int D,
               float eps, hipStream_t s) {
  dim3 grid((M + 3) / 4), block(256);
#define L(NV) hipLaunchKernelGGL((norm_kernel<T, RMS, OUT_F32, NV>), grid, block, 0, s, x, ldx, gamma, beta, y, ldy, M, D, eps)
  if (D <= 512) L(2);
  else if (D <= 1280) {
    if ((int64_t)M * D * 4 >= ((int64_t)64 << 20))           // a stream larger than any cache level (measured: 7.24 -> 6.59 ms per pass)
      hipLaunchKernelGGL((norm_kernel<T, RMS, OUT_F32, 5, true>), grid, block, 0, s, x, ldx, gamma, beta, y, ldy, M, D, eps);
    else L(5);
  }
  else if (D <= 2048) L(8);
  else L(16);
#undef L
}

}  // namespace

namespace {
// (sum x, sum x^2) per row and 64-column slice (the residual GEMM's epilogue, gemm.h) -> (mean, rstd) per row; the slices are added in
// double, so the only rounding is that of the 64-element fp32 partial sums
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int n_slices, float inv_d, float eps, float* __restrict__ stat, int M) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= M) return;
  const float2* p = reinterpret_cast<const float2*>(part) + (int64_t)row * n_slices;
  double s1 = 0.0, s2 = 0.0;
  for (int t = 0; t < n_slices; ++t) { const float2 v = p[t]; s1 += (double)v.x; s2 += (double)v.y; }
  const double mean = s1 * (double)inv_d;
  double var = s2 * (double)inv_d - mean * mean;
  var = var > 0.0 ? var : 0.0;
  *reinterpret_cast<float2*>(stat + 2 * (int64_t)row) = make_float2((float)mean, rsqrtf((float)var + eps));
}
}  // namespace

int mia_ln_finalize_launch(const float* part, int n_slices, int D, float eps, float* stat, int M, hipStream_t s) {
  if (!part || !stat || n_slices <= 0 || D <= 0 || M <= 0) return -1;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part, n_slices, 1.0f / (float)D, eps, stat, M);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

const char* mia_norm_check(int M, int D, int64_t ldx, int64_t ldy) {
  if (M <= 0 || D <= 0) return "norm: M and D must be > 0";
  if (D % 4 || ldx % 4 || ldy % 4) return "norm: D, ldx, ldy must be multiples of 4";
  if (D > 4096) return "norm: D must be <= 4096";
  return nullptr;
}

int mia_norm_launch(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int M, int D,
                    float eps, bool rms, int out_dtype, hipStream_t s) {
#define L(T, R, F) launch_nv<T, R, F>(x, ldx, gamma, beta, y, ldy, M, D, eps, s)
  if (out_dtype == MIA_F32) { if (rms) L(BF16, true, true); else L(BF16, false, true); }
  else if (out_dtype == MIA_F16) { if (rms) L(F16, true, false); else L(F16, false, false); }
  else { if (rms) L(BF16, true, false); else L(BF16, false, false); }
#undef L
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
