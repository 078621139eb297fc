// launch_floor.hip -- what one dependent kernel costs on MI355X, by submission method.
//   hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip && ./launch_floor
// A chain of N dependent kernels (each reads what the previous wrote) is submitted (a) as N stream launches from a tight host loop,
// (b) as one captured hipGraph, for three grid sizes and two kernel bodies (empty / one dependent HBM round trip per workgroup).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 0.f; }

// every workgroup reads one line written by the previous kernel (another XCD wrote it: an L2 miss) and writes one
__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out, int n) {
  int i = (blockIdx.x * 977 + 131) % n;
  float v = in[(size_t)i * 64 + threadIdx.x % 64];
  out[(size_t)blockIdx.x * 64 + threadIdx.x % 64] = v + 1.f;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 2000;
  float *a, *b; CK(hipMalloc(&a, 4096 * 64 * 4)); CK(hipMalloc(&b, 4096 * 64 * 4));
  CK(hipMemset(a, 0, 4096 * 64 * 4)); CK(hipMemset(b, 0, 4096 * 64 * 4));
  int grids[3] = {1, 256, 2048};
  for (int body = 0; body < 2; body++)
    for (int gi = 0; gi < 3; gi++) {
      int g = grids[gi];
      auto chain = [&](hipStream_t st) {
        for (int i = 0; i < N; i++) {
          if (body == 0) k_empty<<<g, 256, 0, st>>>(a);
          else k_touch<<<g, 256, 0, st>>>((i & 1) ? b : a, (i & 1) ? a : b, g);
        }
      };
      // (a) stream launches
      chain(s); CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      chain(s);
      auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      auto t2 = std::chrono::steady_clock::now();
      double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
      double tot_us = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
      // (b) graph
      hipGraph_t gr; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      chain(s);
      CK(hipStreamEndCapture(s, &gr));
      CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
      auto t3 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
      auto t4 = std::chrono::steady_clock::now();
      double graph_us = std::chrono::duration<double, std::micro>(t4 - t3).count() / N;
      printf("body=%s grid=%4d : stream %.2f us/kernel (host submit %.2f), graph %.2f us/kernel\n", body ? "touch" : "empty", g, tot_us, host_us, graph_us);
      CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
    }
  return 0;
}
