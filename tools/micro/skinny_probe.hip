// skinny_probe.hip -- what limits one kernel of the decode chain?  The library's own skinny-GEMM kernels (included as source) are
// replayed from a hipGraph over a pool of distinct weight matrices larger than the 256 MB MALL (every launch streams HBM-cold weights,
// as in the real step), in several tile / wave / pipeline forms per shape.  Prints us per launch (graph replay, boundary included).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mlx-swift-audio_amd/csrc -o skinny_probe skinny_probe.hip
#include "../../mlx-swift-audio_amd/csrc/decode_kernels.hip"

#include <chrono>
#include <cstdio>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_nop(float* p) { if (p == nullptr) p[0] = 0.f; }

// floor: stream `bytes` with fully coalesced 16-byte loads (1 KB per wave instruction), NL loads in flight per lane, one store per wave
template <int NL, bool NT_>
__global__ __launch_bounds__(256) void k_stream(const uint16_t* __restrict__ w, float* __restrict__ out, int per_wg_chunks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const s16x8* p = reinterpret_cast<const s16x8*>(w) + ((size_t)blockIdx.x * 4 + wave) * per_wg_chunks * 64 + lane;
  s16x8 v[NL];
  int acc = 0;
  for (int c = 0; c < per_wg_chunks; c += NL) {
#pragma unroll
    for (int u = 0; u < NL; ++u) v[u] = NT_ ? __builtin_nontemporal_load(p + (size_t)(c + u) * 64) : p[(size_t)(c + u) * 64];
#pragma unroll
    for (int u = 0; u < NL; ++u) acc += v[u][0] + v[u][7];
  }
  if (acc == 0x7fffffff) out[blockIdx.x] = 1.f;
}

// dec_skinny_flat with the weights addressed in MFMA-fragment order (tile, k-step, lane): one contiguous 1 KB per wave instruction.
// Timing only (the pool holds constant bytes, so the values do not matter).
// ABL bit 0: no activation loads (a constant), bit 1: no weight loads, bit 2: no epilogue stores, bit 3: no cross-wave reduction,
// bit 4: activations in fragment order too (1 KB contiguous per wave instruction)
template <typename T, int MODE, int NT, int NSTEP, int NW, int ABL = 0>
__global__ __launch_bounds__(64 * NW) void probe_flat_frag(SkinnyArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * (16 * NT);
  const int split = blockIdx.y;
  const int m0 = blockIdx.z * 32;
  constexpr int Kc = 32 * NSTEP;
  const int kbeg = (split * NW + wave) * Kc;
  const int r = lane & 15, c = lane >> 4;
  const int ksteps = a.K / 32;
  const uint16_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = a.W + (((size_t)(blockIdx.x * NT + t) * ksteps + kbeg / 32) * 64 + lane) * 8;
  int am0 = m0 + r; am0 = am0 < a.M ? am0 : a.M - 1;
  int am1 = m0 + 16 + r; am1 = am1 < a.M ? am1 : a.M - 1;
  const uint16_t* ap0 = a.A + (int64_t)am0 * a.lda + kbeg + 8 * c;
  const uint16_t* ap1 = a.A + (int64_t)am1 * a.lda + kbeg + 8 * c;
  s16x8 fw[NSTEP][NT], fa0[NSTEP], fa1[NSTEP];
#pragma unroll
  for (int u = 0; u < NSTEP; ++u) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      if (ABL & 2) fw[u][n] = (s16x8){(short)lane, 1, 2, 3, 4, 5, 6, (short)u};
      else fw[u][n] = __builtin_nontemporal_load(reinterpret_cast<const s16x8*>(wp[n] + 512 * u));
    }
    if (ABL & 1) { fa0[u] = (s16x8){(short)lane, 1, 2, 3, 4, 5, 6, (short)u}; fa1[u] = fa0[u]; }
    else if (ABL & 16) {
      fa0[u] = *reinterpret_cast<const s16x8*>(a.A + ((size_t)((kbeg / 32 + u) * 2 + 0) * 64 + lane) * 8);
      fa1[u] = *reinterpret_cast<const s16x8*>(a.A + ((size_t)((kbeg / 32 + u) * 2 + 1) * 64 + lane) * 8);
    } else {
      fa0[u] = *reinterpret_cast<const s16x8*>(ap0 + 32 * u);
      fa1[u] = *reinterpret_cast<const s16x8*>(ap1 + 32 * u);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int u = 0; u < NSTEP; ++u)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      acc[n][0] = T::mfma16(fw[u][n], fa0[u], acc[n][0]);
      acc[n][1] = T::mfma16(fw[u][n], fa1[u], acc[n][1]);
    }
  if (ABL & 8) { if (wave > 0) return; }
  else if (!skinny_wave_reduce<NT, NW>(acc, wave, lane)) return;
  if ((ABL & 4) && acc[0][0][0] != 12345.f) return;
  skinny_epilogue<T, MODE, NT>(a, acc, n0, m0, split, lane);
}

static hipStream_t g_s;

// time `reps` launches produced by `launch(i)` (i selects the weight matrix) as one graph
static double time_graph(int reps, const std::function<void(int)>& launch) {
  hipGraph_t gr; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(g_s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipStreamEndCapture(g_s, &gr));
  CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, g_s)); CK(hipStreamSynchronize(g_s));
  double best = 1e30;
  for (int r = 0; r < 3; ++r) {
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ex, g_s)); CK(hipStreamSynchronize(g_s));
    auto t1 = std::chrono::steady_clock::now();
    best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
  }
  CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
  return best;
}

int main() {
  CK(hipStreamCreate(&g_s));
  const int M = 32, D = 1280;
  const size_t pool_bytes = (size_t)640 << 20;              // 640 MB of weights, cycled
  uint16_t* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0x11, pool_bytes));
  uint16_t* A; CK(hipMalloc(&A, (size_t)M * 8192 * 2)); CK(hipMemset(A, 0x22, (size_t)M * 8192 * 2));
  float* bias; CK(hipMalloc(&bias, 70000 * 4)); CK(hipMemset(bias, 0, 70000 * 4));
  void* out; CK(hipMalloc(&out, (size_t)4 * M * 52000 * 4));
  const int reps = 200;
  auto args = [&](int i, int N, int K, int S, int act) {
    const size_t wbytes = (size_t)N * K * 2;
    const int n_mats = (int)(pool_bytes / wbytes);
    SkinnyArgs a{A, K, pool + (size_t)(i % n_mats) * (wbytes / 2), bias, out, N, nullptr, nullptr, nullptr, M, N, K, S, act, D, 20, 448};
    return a;
  };
  printf("nop 320x256: %.2f us\n", time_graph(reps, [&](int) { k_nop<<<320, 256, 0, g_s>>>((float*)out); }));
#define RUN(label, N_, K_, S_, act_, KERNEL, gx, threads)                                                              \
  printf("%-58s %.2f us\n", label, time_graph(reps, [&](int i) {                                                       \
    SkinnyArgs a = args(i, N_, K_, S_, act_);                                                                          \
    hipLaunchKernelGGL(KERNEL, dim3(gx, S_, 1), dim3(threads), 0, g_s, a); }))
  // ---- floor: the same bytes streamed with fully coalesced loads; total = grid x 4 waves x chunks x 1 KB (host-checked against the pool)
  auto stream = [&](const char* label, auto kernel, int grid, int chunks) {
    const size_t bytes = (size_t)grid * 4 * chunks * 1024;
    const int n_mats = (int)(pool_bytes / bytes);
    if (n_mats < 1) { printf("%s: does not fit the pool\n", label); exit(1); }
    printf("%-58s %.2f us  (%.1f MB)\n", label, time_graph(reps, [&](int i) {
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, g_s, pool + (size_t)(i % n_mats) * (bytes / 2), (float*)out, chunks); }), bytes / 1e6);
  };
  stream("stream 320 WG x 10 KB/wave, 10 loads in flight", k_stream<10, false>, 320, 10);
  stream("stream 320 WG x 10 KB/wave, 10 in flight, nt", k_stream<10, true>, 320, 10);
  stream("stream 320 WG x 10 KB/wave, 5 loads in flight", k_stream<5, false>, 320, 10);
  stream("stream 640 WG x 5 KB/wave, 5 loads in flight", k_stream<5, false>, 640, 5);
  stream("stream 160 WG x 5 KB/wave, 5 loads in flight", k_stream<5, false>, 160, 5);
  stream("stream 3242 WG x 10 KB/wave, 10 loads in flight", k_stream<10, false>, 3242, 10);
  stream("stream 3242 WG x 10 KB/wave, 10 in flight, nt", k_stream<10, true>, 3242, 10);
  stream("stream 811 WG x 40 KB/wave, 8 loads in flight", k_stream<8, false>, 811, 40);
  stream("stream 811 WG x 40 KB/wave, 8 in flight, nt", k_stream<8, true>, 811, 40);
  RUN("mlp1 FRAG  NT1 NS10 NW4 grid 320", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4>), 320, 256);
  RUN("mlp1 FRAG  ... no A loads", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 1>), 320, 256);
  RUN("mlp1 FRAG  ... no W loads", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 2>), 320, 256);
  RUN("mlp1 FRAG  ... no A, no W loads", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 3>), 320, 256);
  RUN("mlp1 FRAG  ... no epilogue", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 4>), 320, 256);
  RUN("mlp1 FRAG  ... no reduce", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 8>), 320, 256);
  RUN("mlp1 FRAG  ... A in fragment order", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 16>), 320, 256);
  RUN("mlp1 FRAG  ... A frag, no epilogue", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 1, 10, 4, 20>), 320, 256);
  RUN("mlp1 FRAG  NT2 ... A in fragment order", 5120, 1280, 1, MIA_ACT_GELU, (probe_flat_frag<BF16, SK_OUT16, 2, 10, 4, 16>), 160, 256);
  RUN("mlp2 FRAG  NT1 NS10 NW4 S4 grid 80x4", 1280, 5120, 4, MIA_ACT_NONE, (probe_flat_frag<BF16, SK_PARTIAL, 1, 10, 4>), 80, 256);
  RUN("mlp2 FRAG  ... A in fragment order", 1280, 5120, 4, MIA_ACT_NONE, (probe_flat_frag<BF16, SK_PARTIAL, 1, 10, 4, 16>), 80, 256);
  RUN("oproj FRAG NT1 NS5 NW4 S2 grid 80x2", 1280, 1280, 2, MIA_ACT_NONE, (probe_flat_frag<BF16, SK_PARTIAL, 1, 5, 4>), 80, 256);
  RUN("oproj FRAG ... A in fragment order", 1280, 1280, 2, MIA_ACT_NONE, (probe_flat_frag<BF16, SK_PARTIAL, 1, 5, 4, 16>), 80, 256);
  RUN("oproj FRAG ... no A, no W", 1280, 1280, 2, MIA_ACT_NONE, (probe_flat_frag<BF16, SK_PARTIAL, 1, 5, 4, 3>), 80, 256);
  // ---- mlp1 shape: N 5120, K 1280, GELU, 16-bit out (13.1 MB of weights)
  RUN("mlp1 ring  NT1 KB2 NW4  grid 320", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_gemm<BF16, SK_OUT16, 1, 2, 4>), 320, 256);
  RUN("mlp1 flat  NT1 NS10 NW4 grid 320", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_flat<BF16, SK_OUT16, 1, 10, 4>), 320, 256);
  RUN("mlp1 flat  NT1 NS10 NW4 grid 320 no-GELU", 5120, 1280, 1, MIA_ACT_NONE, (dec_skinny_flat<BF16, SK_OUT16, 1, 10, 4>), 320, 256);
  RUN("mlp1 flat  NT2 NS10 NW4 grid 160", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_flat<BF16, SK_OUT16, 2, 10, 4>), 160, 256);
  RUN("mlp1 flat  NT1 NS5  NW8 grid 320", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_flat<BF16, SK_OUT16, 1, 5, 8>), 320, 512);
  RUN("mlp1 flat  NT2 NS5  NW8 grid 160", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_flat<BF16, SK_OUT16, 2, 5, 8>), 160, 512);
  RUN("mlp1 flat  NT4 NS5  NW8 grid 80", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_flat<BF16, SK_OUT16, 4, 5, 8>), 80, 512);
  RUN("mlp1 ring  NT1 KB4 NW1  grid 320 x 64 thr", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_gemm<BF16, SK_OUT16, 1, 4, 1>), 320, 64);
  RUN("mlp1 ring  NT2 KB2 NW2  grid 160 x 128 thr", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_gemm<BF16, SK_OUT16, 2, 2, 2>), 160, 128);
  // ---- o-proj shape: N 1280, K 1280, split 2, fp32 partials (3.3 MB)
  RUN("oproj ring NT1 KB2 NW4 S2 grid 80x2", 1280, 1280, 2, MIA_ACT_NONE, (dec_skinny_gemm<BF16, SK_PARTIAL, 1, 2, 4>), 80, 256);
  RUN("oproj flat NT1 NS5 NW4 S2 grid 80x2", 1280, 1280, 2, MIA_ACT_NONE, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 5, 4>), 80, 256);
  RUN("oproj flat NT1 NS5 NW8 S1 grid 80", 1280, 1280, 1, MIA_ACT_NONE, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 5, 8>), 80, 512);
  // ---- mlp2 shape: N 1280, K 5120, split 4 (13.1 MB)
  RUN("mlp2 ring  NT1 KB2 NW4 S4 grid 80x4", 1280, 5120, 4, MIA_ACT_NONE, (dec_skinny_gemm<BF16, SK_PARTIAL, 1, 2, 4>), 80, 256);
  RUN("mlp2 flat  NT1 NS10 NW4 S4 grid 80x4", 1280, 5120, 4, MIA_ACT_NONE, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 10, 4>), 80, 256);
  RUN("mlp2 flat  NT1 NS5 NW8 S4 grid 80x4", 1280, 5120, 4, MIA_ACT_NONE, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 5, 8>), 80, 512);
  RUN("mlp2 flat  NT1 NS5 NW4 S8 grid 80x8", 1280, 5120, 8, MIA_ACT_NONE, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 5, 4>), 80, 256);
  // ---- logits shape: N 51866, K 1280, fp32 out (132.8 MB), fragment-order operands (the pool's bytes stand in for both layouts)
  RUN("logits fring NT4 KB2 NW1 grid 811 x 64 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 4, 2, 1>), 811, 64);
  RUN("logits fring NT4 KB4 NW1 grid 811 x 64 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 4, 4, 1>), 811, 64);
  RUN("logits fring NT2 KB4 NW1 grid 1621 x 64 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 2, 4, 1>), 1621, 64);
  RUN("logits fring NT2 KB2 NW2 grid 1621 x 128 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 2, 2, 2>), 1621, 128);
  RUN("logits fring NT4 KB2 NW2 grid 811 x 128 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 4, 2, 2>), 811, 128);
  RUN("logits fring NT1 KB4 NW1 grid 3242 x 64 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 1, 4, 1>), 3242, 64);
  RUN("logits fflat NT1 NS10 NW4 grid 3242 x 256 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fflat<BF16, SK_OUTF32, 1, 10, 4>), 3242, 256);
  RUN("logits fflat NT2 NS10 NW4 grid 1621 x 256 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fflat<BF16, SK_OUTF32, 2, 10, 4>), 1621, 256);
  RUN("logits fring NT8 KB1 NW1 grid 406 x 64 thr", 51866, 1280, 1, MIA_ACT_NONE, (dec_skinny_fring<BF16, SK_OUTF32, 8, 1, 1>), 406, 64);
  // library kernels on the step's other shapes
  RUN("mlp1 fflat NT1 NS10 NW4 grid 320 (library)", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_fflat<BF16, SK_OUT16, 1, 10, 4>), 320, 256);
  RUN("mlp1 fflat NT2 NS10 NW4 grid 160 (library)", 5120, 1280, 1, MIA_ACT_GELU, (dec_skinny_fflat<BF16, SK_OUT16, 2, 10, 4>), 160, 256);
  RUN("mlp2 fflat NT1 NS10 NW4 S4 grid 80x4 (library)", 1280, 5120, 4, MIA_ACT_NONE, (dec_skinny_fflat<BF16, SK_PARTIAL, 1, 10, 4>), 80, 256);
  RUN("mlp2 fflat NT1 NS5 NW4 S8 grid 80x8 (library)", 1280, 5120, 8, MIA_ACT_NONE, (dec_skinny_fflat<BF16, SK_PARTIAL, 1, 5, 4>), 80, 256);
  RUN("oproj fflat NT1 NS5 NW4 S2 grid 80x2 (library)", 1280, 1280, 2, MIA_ACT_NONE, (dec_skinny_fflat<BF16, SK_PARTIAL, 1, 5, 4>), 80, 256);
  // ---- Orpheus-3B step shapes (row-major activations; weights row-major vs fragment order), 1 and 32 sequences
  for (int rows : {1, 32}) {
    for (int wf = 0; wf < 2; ++wf) {
      char lab[96];
#define RUNL(name, N_, K_, S_, KERNEL, gx, threads)                                                                    \
      snprintf(lab, sizeof lab, "%-28s M=%2d %s", name, rows, wf ? "W fragment order" : "W row-major");                \
      printf("%-58s %.2f us\n", lab, time_graph(reps, [&](int i) {                                                      \
        SkinnyArgs a = args(i, N_, K_, S_, MIA_ACT_NONE); a.M = rows; a.w_frag = wf;                                    \
        hipLaunchKernelGGL(KERNEL, dim3(gx, S_, 1), dim3(threads), 0, g_s, a); }))
      RUNL("qkv  5120x3072 S4 NW1 KB4", 5120, 3072, 4, (dec_skinny_gemm<BF16, SK_PARTIAL, 1, 4, 1>), 320, 64);
      RUNL("qkv  5120x3072 S4 NW4 flat6", 5120, 3072, 4, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 6, 4>), 320, 256);
      RUNL("o    3072x3072 S4 NW4 flat6", 3072, 3072, 4, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 6, 4>), 192, 256);
      RUNL("gu  16384x3072 S1 NW4 ring", 16384, 3072, 1, (dec_skinny_gemm<BF16, SK_SWIGLU, 1, 2, 4>), 1024, 256);
      RUNL("down 3072x8192 S8 NW1 KB4", 3072, 8192, 8, (dec_skinny_gemm<BF16, SK_PARTIAL, 1, 4, 1>), 192, 64);
      RUNL("down 3072x8192 S8 NW4 flat8", 3072, 8192, 8, (dec_skinny_flat<BF16, SK_PARTIAL, 1, 8, 4>), 192, 256);
      RUNL("head 65536x3072 NT4 NW1 ring", 65536, 3072, 1, (dec_skinny_gemm<BF16, SK_OUTF32, 4, 2, 1>), 1024, 64);
    }
  }
  return 0;
}
