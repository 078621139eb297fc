// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4_f32 with 1, 2 or 4 independent accumulators per wave,
// and 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// operands rotate through 8 distinct registers loaded from memory (as in a real tile loop), accumulators: NACC independent chains
template <int NACC>
__global__ void k32v(float* out, const float* in, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (8 + i)]; }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
void run(const char* name, F launch, double flop_per_mfma, int nacc, int waves_per_cu) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * nacc * 256 * waves_per_cu;
  printf("%-28s nacc=%d waves/CU=%d : %8.3f ms  %7.1f TFLOP/s  (%.0f ns per MFMA per wave)\n", name, nacc, waves_per_cu, ms,
         mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * nacc));
  hipFree(out);
}

int main() {
#define R32(N, W) run("mfma_f32_32x32x2_f32", [&](float* o, int it) { hipLaunchKernelGGL(k32<N>, dim3(256), dim3(64 * W), 0, 0, o, it, 1.0f, 2.0f); }, 4096.0, N, W)
#define R16(N, W) run("mfma_f32_16x16x4_f32", [&](float* o, int it) { hipLaunchKernelGGL(k16<N>, dim3(256), dim3(64 * W), 0, 0, o, it, 1.0f, 2.0f); }, 2048.0, N, W)
  R32(1, 4); R32(2, 4); R32(4, 4); R32(1, 8); R32(4, 8); R32(1, 16);
  {
    float* in; hipMalloc(&in, 64 * 16 * 4 * 16); hipMemset(in, 0, 64 * 16 * 4 * 16);
#define RV(N, W) run("32x32x2 rotating operands", [&](float* o, int it) { hipLaunchKernelGGL(k32v<N>, dim3(256), dim3(64 * W), 0, 0, o, in, it); }, 4096.0, N, W)
    RV(1, 4); RV(2, 4); RV(1, 8); RV(2, 8);
  }
  R16(1, 4); R16(2, 4); R16(4, 4); R16(4, 8);
  return 0;
}
