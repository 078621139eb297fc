// lds_atomic_rate.hip -- how fast are LDS atomics on gfx950?  One workgroup of 1024 threads (the sampler's shape); every lane issues
// N atomics of one kind to (a) a lane-private word in its own bank, (b) one word shared by 8 lanes, (c) one word shared by the wave.
// Prints lane-operations per clock for ds_add_f32 / ds_add_u32 / ds_add_u64.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T, int SHARE>
__global__ __launch_bounds__(1024) void k(int n, long long* out, T* sink) {
  __shared__ T h[16 * 65 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16 * 65 * 8; i += 1024) h[i] = T(0);
  __syncthreads();
  const int base = (lane / SHARE);                  // SHARE lanes share a word; distinct words sit in distinct banks
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) atomicAdd(&h[base + 65 * (i & 7)], T(1));   // 8 rotating rows so consecutive atomics do not chain
  __syncthreads();
  const long long t1 = wall_clock64();
  if (tid == 0) { out[0] = t1 - t0; }
  sink[tid] = h[tid];
}

template <typename T, int SHARE>
void run(const char* name) {
  long long* d; T* sink; long long hst;
  hipMalloc(&d, 8); hipMalloc(&sink, 1024 * sizeof(T));
  const int n = 2048;
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL((k<T, SHARE>), dim3(1), dim3(1024), 0, 0, n, d, sink); hipDeviceSynchronize(); }
  hipMemcpy(&hst, d, 8, hipMemcpyDeviceToHost);
  const double us = hst / 100.0;                    // wall_clock64: 100 MHz
  printf("%-28s share=%2d  %8.1f us for %d x 1024 lane-ops  -> %.2f ns per wave-instruction, %.3f lane-ops/ns\n", name, SHARE, us, n, us * 1e3 / (n * 16.0), n * 1024.0 / (us * 1e3));
  hipFree(d); hipFree(sink);
}

int main() {
  run<float, 1>("ds_add_f32"); run<float, 8>("ds_add_f32"); run<float, 64>("ds_add_f32");
  run<unsigned, 1>("ds_add_u32"); run<unsigned, 8>("ds_add_u32"); run<unsigned, 64>("ds_add_u32");
  run<unsigned long long, 1>("ds_add_u64"); run<unsigned long long, 8>("ds_add_u64"); run<unsigned long long, 64>("ds_add_u64");
  return 0;
}
