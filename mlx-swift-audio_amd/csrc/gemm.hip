// gemm.hip -- dense "NT" contraction on the gfx950 matrix cores: C[M,N] = epi(A[M,K] . W[N,K]^T).
//
// This is the only place the encoder's true dense contractions run:
// Linear layers (MLXNN Linear: y = x W^T + b, weight [out,in]) and the two Conv1d layers of
// AudioEncoder.swift:47-48, which are expressed as GEMMs over *overlapping* row windows of the channels-last
// input (row stride = conv stride * Cin, K = 3 * Cin) -- no im2col buffer exists.
//
// Structure (CDNA4, wave64): 128x128x64 tile, 256 threads = 2x2 waves, each wave 64x64 as 4x4
// v_mfma_f32_16x16x32 tiles with fp32 accumulation.  Operands are staged HBM -> registers -> LDS
// (issue-early / write-late, one barrier per K-step, double-buffered LDS), rows of 128 B with the 16-B chunk
// index XOR-swizzled by (row & 7) so every ds_read_b128 of a fragment is bank-conflict free.  The MFMA is
// issued with W as the "A" operand, so each lane's 4 accumulator registers run along N (contiguous in C):
// the epilogue emits 8-byte (16-bit C) or 16-byte (fp32 C) stores.
// Workgroup ids are remapped so that the blocks sharing an XCD (and its private L2) walk neighbouring tiles.
#include <mutex>
#include <type_traits>

#include "gemm.h"
#include "mia_device.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KB per operand tile

template <typename T>
__device__ __forceinline__ float apply_act(float v, int act) {
  return act == MIA_ACT_GELU ? gelu_erf(v) : v;
}

// Tile order inside the contiguous block-id range an XCD receives: groups of GROUP_M row-panels walked column-major, so the
// ~32 workgroups resident on one XCD cover ~8 row-panels x 4 column-panels and every A / W K-slice pulled into that XCD's L2 is
// shared by 4..8 of them (row-major order re-fetched the whole W matrix for every row-panel: 9x the algorithmic bytes in
// rocprof FETCH_SIZE for the fused QKV GEMM).
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int group_sz = GROUP_M * tiles_n;
  const int gid = bid / group_sz;
  const int first_m = gid * GROUP_M;
  const int gm = min(tiles_m - first_m, GROUP_M);
  const int r = bid - gid * group_sz;
  tm = first_m + r % gm;
  tn = r / gm;
}

__device__ __forceinline__ void glds16(const void* g, void* l) {   // async 16 B/lane HBM -> LDS (wave-uniform LDS base + lane*16)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ f32x4 load_bias4(const GemmArgs& g, int n) {
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) if (n + j < g.N) b[j] = g.bias[n + j];
  }
  return b;
}

// One lane's 4 consecutive output columns (n .. n+3) of row m: bias, activation, residual, store in the epilogue's layout.
template <typename T, bool OUT_F32, int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int bz, int m, int n, const f32x4& a, const f32x4& bias4, bool vec_ok) {
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = a[j] + bias4[j];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = apply_act<T>(v[j], g.act);
  if (g.R) {
    const float* r = g.R + (int64_t)bz * g.strideR + (int64_t)m * g.ldr + n;
    if (vec_ok) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(r);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += rv[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (n + j < g.N) v[j] += r[j];
    }
  }
  if (EPI == MIA_EPI_STD) {
    if (OUT_F32) {
      float* c = reinterpret_cast<float*>(g.C) + (int64_t)bz * g.strideC + (int64_t)m * g.ldc + n;
      if (vec_ok) *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < g.N) c[j] = v[j];
      }
    } else {
      uint16_t* c = reinterpret_cast<uint16_t*>(g.C) + (int64_t)bz * g.strideC + (int64_t)m * g.ldc + n;
      if (vec_ok) *reinterpret_cast<u32x2*>(c) = (u32x2){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j < g.N) c[j] = T::from_f32(v[j]);
      }
    }
  } else if (EPI == MIA_EPI_QKV_VT) {
    // columns [0, 2D): q|k rows, row stride ldc.  columns [2D, 3D): V, stored transposed per head:
    // vt[((b*H + h)*64 + d) * Tpad + t]   (m = b*T + t, n - 2D = h*64 + d)
    const int D2 = 2 * g.H * 64;
    if (n < D2) {
      uint16_t* c = reinterpret_cast<uint16_t*>(g.C) + (int64_t)m * g.ldc + n;
      *reinterpret_cast<u32x2*>(c) = (u32x2){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
    } else {
      const int b = m / g.T, t = m - b * g.T;
      const int hd = n - D2;   // h*64 + d
      uint16_t* vt = reinterpret_cast<uint16_t*>(g.C2) + ((int64_t)b * g.H * 64 + hd) * g.Tpad + t;
#pragma unroll
      for (int j = 0; j < 4; ++j) vt[(int64_t)j * g.Tpad] = T::from_f32(v[j]);
    }
  } else {  // MIA_EPI_HEADMAJOR: out[((b*H + h)*T + t)*64 + d], n = h*64 + d (4 consecutive n share a head)
    const int b = m / g.T, t = m - b * g.T;
    const int h = n >> 6, d = n & 63;
    uint16_t* c = reinterpret_cast<uint16_t*>(g.C) + (((int64_t)b * g.H + h) * g.T + t) * 64 + d;
    *reinterpret_cast<u32x2*>(c) = (u32x2){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
  }
}

// STAGE 0: HBM -> registers -> LDS (issue early, write late).  STAGE 1: LDS-DMA (global_load_lds_dwordx4): the LDS image is
// lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address; no staging registers, no ds_write pass.
template <typename T, bool OUT_F32, int EPI, int STAGE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // [2 buffers][A|W]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware tile mapping (bijective form; MI355X deals consecutive block ids round-robin to 8 XCDs)
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int tm, tn;
  tile_of(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int bz = blockIdx.z;

  const uint16_t* __restrict__ A = reinterpret_cast<const uint16_t*>(g.A) + (int64_t)bz * g.strideA;
  const uint16_t* __restrict__ W = reinterpret_cast<const uint16_t*>(g.W);

  // ---- staging: each thread moves 4 x 16 B of A and 4 x 16 B of W per K-step.
  // Piece i of this wave covers tile rows 32*wave + 8*i + (lane>>3); lane&7 is the 16-B position inside the 128-B row,
  // which holds logical chunk (lane&7) ^ (row&7).
  const int s_pos = lane & 7;
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
  int s_dst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wave + 8 * i + (lane >> 3);
    const int chk = s_pos ^ (row & 7);
    int am = m0 + row; am = am < g.M ? am : g.M - 1;      // clamp: tail rows are computed but never stored
    int wn = n0 + row; wn = wn < g.N ? wn : g.N - 1;
    a_src[i] = A + (int64_t)am * g.lda + chk * 8;
    w_src[i] = W + (int64_t)wn * g.K + chk * 8;
    s_dst[i] = row * 128 + (s_pos << 4);                  // == (32*wave + 8*i)*128 + lane*16: lane-linear per piece
  }
  u32x4 ra[4], rw[4];
  auto load_regs = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(a_src[i] + k0);
      rw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + k0);
    }
  };
  auto write_lds = [&](int buf) {
    char* base = lds + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(base + s_dst[i]) = ra[i];
      *reinterpret_cast<u32x4*>(base + TILE_BYTES + s_dst[i]) = rw[i];
    }
  };
  auto stage_dma = [&](int buf, int k0) {
    char* base = lds + buf * 2 * TILE_BYTES + (32 * wave) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + k0, base + i * 1024);
      glds16(w_src[i] + k0, base + TILE_BYTES + i * 1024);
    }
  };

  // ---- fragment read addresses (row = base + lane&15, chunk = kk*4 + lane>>4)
  const int f_row = lane & 15, f_chk = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  auto compute = [&](int cur) {
    const char* sa = lds + cur * 2 * TILE_BYTES;
    const char* sw = sa + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x8 fa[4], fw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ar = wr * 64 + t * 16 + f_row;
        fa[t] = *reinterpret_cast<const s16x8*>(sa + ar * 128 + (((kk * 4 + f_chk) ^ (ar & 7)) << 4));
        const int wrow = wc * 64 + t * 16 + f_row;
        fw[t] = *reinterpret_cast<const s16x8*>(sw + wrow * 128 + (((kk * 4 + f_chk) ^ (wrow & 7)) << 4));
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = T::mfma16(fw[nt], fa[mt], acc[mt][nt]);  // rows<-n, cols<-m
    }
  };
  int cur = 0;
  if (STAGE == 0) {
    load_regs(0);
    write_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) load_regs((kt + 1) * BK);
      compute(cur);
      if (kt + 1 < nk) write_lds(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    stage_dma(0, 0);
    __syncthreads();                       // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage_dma(cur ^ 1, (kt + 1) * BK);   // the other buffer was last read before the previous barrier
      compute(cur);
      __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: lane holds, per (mt,nt), C[m = ..+lane&15][n = ..+(lane>>4)*4 + 0..3]
  const int e_m = lane & 15, e_n = (lane >> 4) * 4;
  const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && (g.R == nullptr || (g.ldr & 3) == 0);
  f32x4 bias4[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bias4[nt] = load_bias4(g, n0 + wc * 64 + nt * 16 + e_n);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wr * 64 + mt * 16 + e_m;
    if (m >= g.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wc * 64 + nt * 16 + e_n;
      if (n >= g.N) continue;
      epilogue_store<T, OUT_F32, EPI>(g, bz, m, n, acc[mt][nt], bias4[nt], vec_ok);
    }
  }
}

// ---- 256x256x64 tile, 512 threads = 2(M) x 4(N) waves, each wave 128x64 (32 accumulator tiles), one block per CU.
// Twice the arithmetic intensity of the 128^2 tile (31 instead of 62 LDS-DMA bytes per MFMA-cycle per CU), which is
// what the L2 -> LDS path can sustain on 256 CUs.  LDS: 2 stages x (A 32 KB + W 32 KB) = 128 KB, LDS-DMA staged,
// same swizzle; per K-step a wave runs 64 MFMAs in four 64x32 quadrants so only 8 fragments are live at a time.
constexpr int BM2 = 256, BN2 = 256;
constexpr int TILE2_BYTES = BM2 * BK * 2;   // 32 KB per operand tile

template <typename T, bool OUT_F32, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel_256(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 stages][A|W] = 128 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.N + BN2 - 1) / BN2;
  const int tiles_m = (g.M + BM2 - 1) / BM2;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int tm, tn;
  tile_of(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int bz = blockIdx.z;
  const uint16_t* __restrict__ A = reinterpret_cast<const uint16_t*>(g.A) + (int64_t)bz * g.strideA;
  const uint16_t* __restrict__ W = reinterpret_cast<const uint16_t*>(g.W);

  const int s_pos = lane & 7;
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wave + 8 * i + (lane >> 3);
    const int chk = s_pos ^ (row & 7);
    int am = m0 + row; am = am < g.M ? am : g.M - 1;
    int wn = n0 + row; wn = wn < g.N ? wn : g.N - 1;
    a_src[i] = A + (int64_t)am * g.lda + chk * 8;
    w_src[i] = W + (int64_t)wn * g.K + chk * 8;
  }
  auto stage_dma = [&](int buf, int k0) {
    char* base = lds + buf * 2 * TILE2_BYTES + (32 * wave) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + k0, base + i * 1024);
      glds16(w_src[i] + k0, base + TILE2_BYTES + i * 1024);
    }
  };
  const int f_row = lane & 15, f_chk = lane >> 4;
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  // V tiles of the fused QKV GEMM are computed with the MFMA operands swapped (rows <- m): a lane's 4 accumulator registers
  // then run along the token axis, which is the contiguous axis of the transposed V^T destination (8-byte stores instead of
  // four scattered 2-byte ones).
  const bool v_tile = EPI == MIA_EPI_QKV_VT && n0 >= 2 * g.H * 64;
  auto compute = [&](int cur, auto swap_tag) {
    constexpr bool SWAP = decltype(swap_tag)::value;
    const char* sa = lds + cur * 2 * TILE2_BYTES;
    const char* sw = sa + TILE2_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x8 fw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int wrow = wc * 64 + t * 16 + f_row;
        fw[t] = *reinterpret_cast<const s16x8*>(sw + wrow * 128 + (((kk * 4 + f_chk) ^ (wrow & 7)) << 4));
      }
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        s16x8 fa[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ar = wr * 128 + mh * 64 + t * 16 + f_row;
          fa[t] = *reinterpret_cast<const s16x8*>(sa + ar * 128 + (((kk * 4 + f_chk) ^ (ar & 7)) << 4));
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[mh * 4 + mt][nt] = SWAP ? T::mfma16(fa[mt], fw[nt], acc[mh * 4 + mt][nt]) : T::mfma16(fw[nt], fa[mt], acc[mh * 4 + mt][nt]);
      }
    }
  };
  auto main_loop = [&](auto swap_tag) {
    int cur = 0;
    stage_dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage_dma(cur ^ 1, (kt + 1) * BK);
      compute(cur, swap_tag);
      __syncthreads();
      cur ^= 1;
    }
  };
  if (EPI == MIA_EPI_QKV_VT && v_tile) main_loop(std::true_type{}); else main_loop(std::false_type{});
  if (EPI == MIA_EPI_QKV_VT && v_tile) {
    // lane holds C[m = ..+(lane>>4)*4 + 0..3][n = ..+(lane&15)]; T % 4 == 0 keeps the 4 rows inside one clip
    const int D2 = 2 * g.H * 64;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + wr * 128 + mt * 16 + (lane >> 4) * 4;
      if (m >= g.M) continue;
      const int b = m / g.T, t = m - b * g.T;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + (lane & 15);
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
        const f32x4 a = acc[mt][nt];
        uint16_t* vt = reinterpret_cast<uint16_t*>(g.C2) + ((int64_t)b * g.H * 64 + (n - D2)) * g.Tpad + t;
        *reinterpret_cast<u32x2*>(vt) = (u32x2){pack2<T>(a[0] + bias, a[1] + bias), pack2<T>(a[2] + bias, a[3] + bias)};
      }
    }
    return;
  }
  const int e_m = lane & 15, e_n = (lane >> 4) * 4;
  const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && (g.R == nullptr || (g.ldr & 3) == 0);
  f32x4 bias4[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bias4[nt] = load_bias4(g, n0 + wc * 64 + nt * 16 + e_n);
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + wr * 128 + mt * 16 + e_m;
    if (m >= g.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wc * 64 + nt * 16 + e_n;
      if (n >= g.N) continue;
      epilogue_store<T, OUT_F32, EPI>(g, bz, m, n, acc[mt][nt], bias4[nt], vec_ok);
    }
  }
}


// ---- 256x256x64 tile, 8 phases per two K-tiles (cdna_hip_programming.md "The 256^2 8-phase template", rebuilt here).
// 512 threads = 2(M) x 4(N) waves, each wave 128x64 (32 accumulator tiles = 128 registers).  Per K-tile four phases of
// 16 MFMAs (one 64x32 quadrant x K=64); every phase is  {ds_reads | one LDS-DMA item | counted vmcnt} s_barrier
// {lgkmcnt(0) | 16 MFMA} s_barrier.  The two wave groups (waves 0-3 = row half 0, waves 4-7 = row half 1; one of each per SIMD)
// run ONE barrier apart, so while one group's wave owns the SIMD's matrix pipe the other one reads LDS and issues DMA.
//
// LDS (128 KB) = 2 K-tile buffers x 4 items of 16 KB (128 rows x 128 B, 16-B chunk index XOR (row & 7)):
//   AX = rows {0..63, 128..191} of the A tile (the first 64 rows of either group's 128), AY = the other 64 of each,
//   B0 / B1 = W rows 0..127 / 128..255.  Items are issued in the fixed order [AX B0 B1 AY] of tile 0, 1, 2, ... -- one per
//   phase, six ahead of the consumer -- so a counted `s_waitcnt vmcnt(6)` (three items stay in flight) retires exactly the
//   item the NEXT phase reads.  Rules kept (MI355X_MICROARCH.md item 7, guide "Read a staged buffer one phase AFTER the
//   wait that retires it"): wait in phase i -> first read in phase i+1; a region is re-staged >= 2 phases after the phase
//   whose lgkmcnt(0) retired its last read (the groups are one barrier apart).
//   reads : phi1 fa0 (8) + fw[nt 0,1] (4) | phi2 fw[nt 2,3] (4) + fa1[mt 4,5] (4) | phi3 fa1[mt 6,7] (4) | phi4 none
//   issue : phi1 B1(t+1) | phi2 AY(t+1) | phi3 AX(t+2) | phi4 B0(t+2)         waits: phi1 -> AY(t), phi4 -> AX,B0,B1(t+1)
constexpr int ITEM_BYTES = 128 * 128;      // 16 KB
constexpr int KBUF_BYTES = 4 * ITEM_BYTES; // one K-tile: [B0][B1][AX][AY]
constexpr int OFF_B0 = 0, OFF_B1 = ITEM_BYTES, OFF_AX = 2 * ITEM_BYTES, OFF_AY = 3 * ITEM_BYTES;

#define MIA_BAR()                                 \
  do {                                            \
    asm volatile("" ::: "memory");                \
    __builtin_amdgcn_s_barrier();                 \
    asm volatile("" ::: "memory");                \
  } while (0)

__device__ __forceinline__ void wait_items(int allowed) {   // wave-uniform: leave `allowed` items (2 LDS-DMA each) in flight
  if (allowed >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (allowed == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (allowed == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T, bool OUT_F32, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel_8ph(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 K-tile buffers][B0|B1|AX|AY] = 128 KB (the only LDS object)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (g.N + BN2 - 1) / BN2;
  const int tiles_m = (g.M + BM2 - 1) / BM2;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int tm, tn;
  tile_of(bid, tiles_m, tiles_n, tm, tn);
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int bz = blockIdx.z;
  const char* __restrict__ Ab = reinterpret_cast<const char*>(reinterpret_cast<const uint16_t*>(g.A) + (int64_t)bz * g.strideA);
  const char* __restrict__ Wb = reinterpret_cast<const char*>(g.W);

  // ---- LDS-DMA sources: this wave fills LDS rows 16*wave + 8*i + (lane>>3), i = 0, 1, of every item; lane&7 is the 16-B position
  // in the 128-B row, which holds logical chunk (lane&7) ^ (row&7): the swizzle sits on the SOURCE address, the LDS image is lane-linear.
  uint32_t ax_off[2], ay_off[2], b0_off[2], b1_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * wave + 8 * i + (lane >> 3);
    const int chk = (lane & 7) ^ (row & 7);
    const int arow = (row >> 6) * 128 + (row & 63);
    int ax = m0 + arow, ay = m0 + arow + 64, w0 = n0 + row, w1 = n0 + 128 + row;
    ax = ax < g.M ? ax : g.M - 1; ay = ay < g.M ? ay : g.M - 1;     // clamp: tail rows are computed but never stored
    w0 = w0 < g.N ? w0 : g.N - 1; w1 = w1 < g.N ? w1 : g.N - 1;
    ax_off[i] = (uint32_t)(((int64_t)ax * g.lda + chk * 8) * 2);
    ay_off[i] = (uint32_t)(((int64_t)ay * g.lda + chk * 8) * 2);
    b0_off[i] = (uint32_t)(((int64_t)w0 * g.K + chk * 8) * 2);
    b1_off[i] = (uint32_t)(((int64_t)w1 * g.K + chk * 8) * 2);
  }
  const int nk = g.K / BK;
  const int n_items = 4 * nk;
  char* const lds_wave = lds + (16 * wave) * 128;
  // LayerNorm hand-over (gemm.h): the tile's per-row (mean, rstd) and per-column c1 (consumer) or gamma (producer) go to 4 KB of LDS
  // behind the two K-tile buffers by LDS-DMA, issued BEFORE the first stage so that they are the oldest operations of every counted
  // wait: fetched at the head of the epilogue they were a bare memory round trip per tile (1 workgroup per CU: nothing hides it;
  // measured + 30 us on the MLP's first Linear).
  char* const lds_ln = lds + 2 * KBUF_BYTES;            // [256 rows][2] stats | [256] c1 or gamma
  if (EPI == MIA_EPI_STD && (g.ln_stat || g.ln_gamma)) {
    if (g.ln_stat) {
      int m = m0 + 32 * wave + (lane >> 1); m = m < g.M ? m : g.M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.ln_stat + 2 * (int64_t)m + (lane & 1)),
                                       (__attribute__((address_space(3))) void*)(lds_ln + wave * 256), 4, 0, 0);
    }
    if (wave < 4) {
      const float* vsrc = g.ln_stat ? g.ln_c1 : g.ln_gamma;
      int n = n0 + 64 * wave + lane; n = n < g.N ? n : g.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + n),
                                       (__attribute__((address_space(3))) void*)(lds_ln + 2048 + wave * 256), 4, 0, 0);
    }
  }
  // item j = 4*t + {0: AX, 1: B0, 2: B1, 3: AY} of K-tile t
  // (`guard` = false in the steady part of the K loop, where every staged item exists: the range checks and the run-time choice of
  // the vmcnt immediate cost 31 scalar branches per pair of K-tiles, in a loop whose 128 MFMAs are otherwise back to back)
  auto stage_item = [&](int j, bool guard = true) {
    if (guard && j >= n_items) return;
    const int t = j >> 2, which = j & 3;
    char* dst = lds_wave + (t & 1) * KBUF_BYTES;
    const int kb = t * (BK * 2);
    if (which == 0) {
      glds16(Ab + ax_off[0] + kb, dst + OFF_AX); glds16(Ab + ax_off[1] + kb, dst + OFF_AX + 1024);
    } else if (which == 1) {
      glds16(Wb + b0_off[0] + kb, dst + OFF_B0); glds16(Wb + b0_off[1] + kb, dst + OFF_B0 + 1024);
    } else if (which == 2) {
      glds16(Wb + b1_off[0] + kb, dst + OFF_B1); glds16(Wb + b1_off[1] + kb, dst + OFF_B1 + 1024);
    } else {
      glds16(Ab + ay_off[0] + kb, dst + OFF_AY); glds16(Ab + ay_off[1] + kb, dst + OFF_AY + 1024);
    }
  };

  // ---- fragment read offsets: row = base + (lane & 15), logical chunk = kk*4 + (lane >> 4)
  const int f_row = lane & 15, f_chk = lane >> 4;
  int a_rd[2], b_rd[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int sw = ((kk * 4 + f_chk) ^ (f_row & 7)) << 4;
    a_rd[kk] = (wr * 64 + f_row) * 128 + sw;                                   // + OFF_AX / OFF_AY + t*2048
    b_rd[kk] = (wc >> 1) * ITEM_BYTES + ((wc & 1) * 64 + f_row) * 128 + sw;    // + t*2048
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool v_tile = EPI == MIA_EPI_QKV_VT && n0 >= 2 * g.H * 64;   // V tiles: operands swapped, rows <- m (see gemm_nt_kernel_256)

  auto main_loop = [&](auto swap_tag) {
    constexpr bool SWAP = decltype(swap_tag)::value;
    s16x8 fa0[2][4], fa1[2][4], fw[2][4];
    auto mma16 = [&](const s16x8 (&fa)[2][4], int mbase, int nbase) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mbase + mt][nbase + nt] = SWAP ? T::mfma16(fa[kk][mt], fw[kk][nbase + nt], acc[mbase + mt][nbase + nt])
                                               : T::mfma16(fw[kk][nbase + nt], fa[kk][mt], acc[mbase + mt][nbase + nt]);
      __builtin_amdgcn_s_setprio(0);
    };
    auto ktile = [&](int kt, auto par_tag, auto steady_tag) {
      constexpr int P = decltype(par_tag)::value;
      constexpr bool STEADY = decltype(steady_tag)::value;   // items j0 + 6 .. j0 + 9 all exist: no guards, constant waits
      const char* buf = lds + P * KBUF_BYTES;
      const int j0 = 4 * kt;
      // ---------------- phi1
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fw[kk][t] = *reinterpret_cast<const s16x8*>(buf + OFF_B0 + b_rd[kk] + t * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa0[kk][t] = *reinterpret_cast<const s16x8*>(buf + OFF_AX + a_rd[kk] + t * 2048);
      }
      stage_item(j0 + 6, !STEADY);
      if (STEADY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else wait_items(min(j0 + 7, n_items) - (j0 + 4));     // AY(kt) landed (read in phi2)
      MIA_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma16(fa0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      MIA_BAR();
      // ---------------- phi2
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 2; t < 4; ++t) fw[kk][t] = *reinterpret_cast<const s16x8*>(buf + OFF_B0 + b_rd[kk] + t * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fa1[kk][t] = *reinterpret_cast<const s16x8*>(buf + OFF_AY + a_rd[kk] + t * 2048);
      }
      stage_item(j0 + 7, !STEADY);
      MIA_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma16(fa0, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      MIA_BAR();
      // ---------------- phi3
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int t = 2; t < 4; ++t) fa1[kk][t] = *reinterpret_cast<const s16x8*>(buf + OFF_AY + a_rd[kk] + t * 2048);
      }
      stage_item(j0 + 8, !STEADY);
      MIA_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mma16(fa1, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
      MIA_BAR();
      // ---------------- phi4
      stage_item(j0 + 9, !STEADY);
      if (STEADY) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (kt + 1 < nk) wait_items(min(j0 + 10, n_items) - (j0 + 7));   // AX, B0, B1 of tile kt+1 landed (read in the next phi1)
      MIA_BAR();
      __builtin_amdgcn_sched_barrier(0);
      mma16(fa1, 4, 2);
      __builtin_amdgcn_sched_barrier(0);
      MIA_BAR();
    };
    // ---- prologue: tile 0 whole + AX, B0 of tile 1 (items 0..5); items 0..2 must have landed before the first reads
#pragma unroll
    for (int j = 0; j < 6; ++j) stage_item(j);
    wait_items(min(6, n_items) - 3);
    MIA_BAR();
    if (wr == 1) MIA_BAR();          // stagger: the second wave group runs one barrier behind the first
    // steady pairs: both tiles stage items up to 4 (kt + 1) + 9 < n_items, i.e. kt + 1 <= nk - 3
    int kt = 0;
    for (; kt + 4 <= nk; kt += 2) {      // (same-process A/B against the all-guarded loop: +0.7 ... +5.5 % on the four encoder shapes)
      ktile(kt, std::integral_constant<int, 0>{}, std::true_type{});
      ktile(kt + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    for (; kt < nk; kt += 2) {
      ktile(kt, std::integral_constant<int, 0>{}, std::false_type{});
      if (kt + 1 < nk) ktile(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    if (wr == 0) MIA_BAR();
  };
  if (EPI == MIA_EPI_QKV_VT && v_tile) main_loop(std::true_type{}); else main_loop(std::false_type{});

  if (EPI == MIA_EPI_QKV_VT && v_tile) {
    const int D2 = 2 * g.H * 64;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = m0 + wr * 128 + mt * 16 + (lane >> 4) * 4;
      if (m >= g.M) continue;
      const int b = m / g.T, t = m - b * g.T;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wc * 64 + nt * 16 + (lane & 15);
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
        const f32x4 a = acc[mt][nt];
        uint16_t* vt = reinterpret_cast<uint16_t*>(g.C2) + ((int64_t)b * g.H * 64 + (n - D2)) * g.Tpad + t;
        *reinterpret_cast<u32x2*>(vt) = (u32x2){pack2<T>(a[0] + bias, a[1] + bias), pack2<T>(a[2] + bias, a[3] + bias)};
      }
    }
    return;
  }
  const int e_m = lane & 15, e_n = (lane >> 4) * 4;
  const bool vec_ok = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && (g.R == nullptr || (g.ldr & 3) == 0);
  f32x4 bias4[4];
  // the wave's 64 columns all exist and the pointer is 16-byte aligned (one wave-uniform test): four back-to-back 16-byte loads, one
  // L2 round trip.  As guarded scalar loads hipcc emitted a branch per element with `s_waitcnt vmcnt(0)` between the groups -- several
  // dependent round trips at the head of every tile's epilogue.
  if (g.bias && (((uintptr_t)g.bias) & 15) == 0 && n0 + wc * 64 + 64 <= g.N) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias4[nt] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wc * 64 + nt * 16 + e_n);
  } else {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias4[nt] = load_bias4(g, n0 + wc * 64 + nt * 16 + e_n);
  }
  if constexpr (!OUT_F32 && (EPI == MIA_EPI_STD || EPI == MIA_EPI_QKV_VT)) {
    // 16-bit row-major output: the direct store is 32 x 8 B per lane in 32-B row pieces and is store-ISSUE-bound (a third of a
    // K = 1280 tile's time).  Stage the wave's 128 x 64 sub-tile in its own 16 KB of the (now idle) LDS -- 128-B rows, 16-B chunk
    // XOR (row & 7) -- and write it out as 16 x 16 B per lane: whole 128-B row segments, half the store instructions.
    const bool rows16 = g.R == nullptr && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (((uintptr_t)g.C) & 15) == 0 && (g.strideC & 7) == 0 &&
                        (EPI == MIA_EPI_STD || n0 + BN2 <= 2 * g.H * 64);
    if (rows16) {
      char* reg = lds + wave * 16384;
      // LayerNorm hand-over (gemm.h): per-row (mean, rstd) of this lane's 8 rows and c1 of its 16 columns, loaded before the staging
      float st_m[8], st_r[8];
      f32x4 c1v[4];
      const bool ln = EPI == MIA_EPI_STD && g.ln_stat != nullptr;
      if (ln) {            // (from LDS: staged at kernel start)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const float2 mr = *reinterpret_cast<const float2*>(lds_ln + (wr * 128 + mt * 16 + e_m) * 8);
          st_m[mt] = -mr.x; st_r[mt] = mr.y;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) c1v[nt] = *reinterpret_cast<const f32x4*>(lds_ln + 2048 + (wc * 64 + nt * 16 + e_n) * 4);
      }
      // (the activation switch is taken ONCE per tile: tested per value it compiled into a branch around every one of the 128 GELUs)
      auto stage16 = [&](auto gelu_tag, auto ln_tag) {
        constexpr bool GELU = decltype(gelu_tag)::value;
        constexpr bool LN = decltype(ln_tag)::value;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const int row = mt * 16 + e_m;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = LN ? fmaf(st_r[mt], fmaf(st_m[mt], c1v[nt][j], acc[mt][nt][j]), bias4[nt][j]) : acc[mt][nt][j] + bias4[nt][j];
              if (GELU) v[j] = gelu_erf(v[j]);
            }
            const int c16 = nt * 2 + (lane >> 5), half = (lane >> 4) & 1;
            *reinterpret_cast<u32x2*>(reg + row * 128 + ((c16 ^ (row & 7)) << 4) + half * 8) = (u32x2){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
          }
        }
      };
      if (ln) { if (g.act == MIA_ACT_GELU) stage16(std::true_type{}, std::true_type{}); else stage16(std::false_type{}, std::true_type{}); }
      else { if (g.act == MIA_ACT_GELU) stage16(std::true_type{}, std::false_type{}); else stage16(std::false_type{}, std::false_type{}); }
      uint16_t* cb = reinterpret_cast<uint16_t*>(g.C) + (EPI == MIA_EPI_STD ? (int64_t)bz * g.strideC : 0);
      const int c16 = lane & 7;
      const int n = n0 + wc * 64 + c16 * 8;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = 8 * i + (lane >> 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(reg + row * 128 + ((c16 ^ (row & 7)) << 4));
        const int m = m0 + wr * 128 + row;
        if (m < g.M && n < g.N) *reinterpret_cast<u32x4*>(cb + (int64_t)m * g.ldc + n) = v;
      }
      return;
    }
  }
  if constexpr (OUT_F32 && EPI == MIA_EPI_STD) {
    // fp32 output (+ fp32 residual): the direct form reads and writes 64-B row pieces, 16 rows per instruction.  Through the wave's 16 KB of
    // LDS (two halves of 64 rows x 256 B, 16-B chunk XOR (row & 15)) every load / store instruction covers 4 whole 256-B row segments,
    // and the residual is fetched with the same coalesced shape.
    const bool rows32 = (g.N & 3) == 0 && (g.ldc & 3) == 0 && (((uintptr_t)g.C) & 15) == 0 && (g.strideC & 3) == 0 &&
                        (g.R == nullptr || ((g.ldr & 3) == 0 && (((uintptr_t)g.R) & 15) == 0 && (g.strideR & 3) == 0));
    if (rows32) {
      char* reg = lds + wave * 16384;
      float* cb = reinterpret_cast<float*>(g.C) + (int64_t)bz * g.strideC;
      const float* rb = g.R ? g.R + (int64_t)bz * g.strideR : nullptr;
      const int c16 = lane & 15;
      const int n = n0 + wc * 64 + c16 * 4;
      const bool gelu32 = g.act == MIA_ACT_GELU;
      // LayerNorm hand-over (gemm.h): gamma of this lane's 4 columns; the row's 64 columns of this wave sit in the 16 lanes of a DPP row
      const bool ln_out = g.ln_gamma != nullptr;
      f32x4 gm = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ln_out) gm = *reinterpret_cast<const f32x4*>(lds_ln + 2048 + (wc * 64 + c16 * 4) * 4);      // (staged at kernel start)
      const int ln_slices = g.N >> 6, ln_slice = (n0 + wc * 64) >> 6;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        // the half's 16 residual vectors first: unconditional loads (row / column clamped into the tensor, unused lanes discard them),
        // so that they are all in flight while the accumulators are staged -- fetched inside the store loop, under its bounds test,
        // each one was a dependent memory round trip (load, s_waitcnt vmcnt(0), add, store, 16 times per half)
        f32x4 rv[16];
        if (rb) {
          const int nc = n < g.N ? n : 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            int m = m0 + wr * 128 + mh * 64 + 4 * i + (lane >> 4);
            m = m < g.M ? m : g.M - 1;
            rv[i] = *reinterpret_cast<const f32x4*>(rb + (int64_t)m * g.ldr + nc);
          }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row = mt * 16 + e_m;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[mh * 4 + mt][nt][j] + bias4[nt][j];
            if (gelu32) {          // uniform; the encoder's fp32-output GEMMs (out-proj, fc2) carry no activation
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            }
            const int wc16 = nt * 4 + (lane >> 4);
            *reinterpret_cast<f32x4*>(reg + row * 256 + ((wc16 ^ (row & 15)) << 4)) = v;
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = 4 * i + (lane >> 4);
          f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 256 + ((c16 ^ (row & 15)) << 4));
          const int m = m0 + wr * 128 + mh * 64 + row;
          if (rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rv[i][j];
          }
          if (m < g.M && n < g.N) *reinterpret_cast<f32x4*>(cb + (int64_t)m * g.ldc + n) = v;
          if (ln_out) {
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
            float s2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
            s1 += dpp_f32<0xB1>(s1); s2 += dpp_f32<0xB1>(s2);
            s1 += dpp_f32<0x4E>(s1); s2 += dpp_f32<0x4E>(s2);
            s1 += dpp_f32<0x141>(s1); s2 += dpp_f32<0x141>(s2);
            s1 += dpp_f32<0x140>(s1); s2 += dpp_f32<0x140>(s2);
            if (m < g.M && n < g.N) {
              if (c16 == 0) *reinterpret_cast<float2*>(g.ln_part + ((int64_t)m * ln_slices + ln_slice) * 2) = make_float2(s1, s2);
              *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(g.ln_out) + (int64_t)m * g.ln_ld + n) =
                  (u32x2){pack2<T>(v[0] * gm[0], v[1] * gm[1]), pack2<T>(v[2] * gm[2], v[3] * gm[3])};
            }
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = m0 + wr * 128 + mt * 16 + e_m;
    if (m >= g.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wc * 64 + nt * 16 + e_n;
      if (n >= g.N) continue;
      epilogue_store<T, OUT_F32, EPI>(g, bz, m, n, acc[mt][nt], bias4[nt], vec_ok);
    }
  }
}

template <typename T>
int launch_8ph(const GemmArgs& g, hipStream_t s) {
  const int tiles = ((g.M + BM2 - 1) / BM2) * ((g.N + BN2 - 1) / BN2);
  dim3 grid(tiles, 1, g.batch > 0 ? g.batch : 1), block(512);
  const size_t lds_bytes = 2 * KBUF_BYTES + 4096;      // two K-tile buffers + the LayerNorm hand-over vectors
#define L8(F32, E)                                                                                                \
  do {                                                                                                            \
    static std::once_flag once;                                                                                   \
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel_8ph<T, F32, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KBUF_BYTES + 4096); }); \
    hipLaunchKernelGGL((gemm_nt_kernel_8ph<T, F32, E>), grid, block, lds_bytes, s, g);                             \
  } while (0)
  if (g.epi == MIA_EPI_STD) { if (g.out_f32) L8(true, MIA_EPI_STD); else L8(false, MIA_EPI_STD); }
  else if (g.epi == MIA_EPI_QKV_VT) L8(false, MIA_EPI_QKV_VT);
  else L8(false, MIA_EPI_HEADMAJOR);
#undef L8
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T>
int launch_256(const GemmArgs& g, hipStream_t s) {
  const int tiles = ((g.M + BM2 - 1) / BM2) * ((g.N + BN2 - 1) / BN2);
  dim3 grid(tiles, 1, g.batch > 0 ? g.batch : 1), block(512);
  const size_t lds_bytes = 4 * TILE2_BYTES;
#define L256(F32, E)                                                                                              \
  do {                                                                                                            \
    static std::once_flag once;                                                                                   \
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel_256<T, F32, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE2_BYTES); }); \
    hipLaunchKernelGGL((gemm_nt_kernel_256<T, F32, E>), grid, block, lds_bytes, s, g);                             \
  } while (0)
  if (g.epi == MIA_EPI_STD) { if (g.out_f32) L256(true, MIA_EPI_STD); else L256(false, MIA_EPI_STD); }
  else if (g.epi == MIA_EPI_QKV_VT) L256(false, MIA_EPI_QKV_VT);
  else L256(false, MIA_EPI_HEADMAJOR);
#undef L256
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T, int STAGE>
int launch_ts(const GemmArgs& g, hipStream_t s) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  dim3 grid(tiles, 1, g.batch > 0 ? g.batch : 1), block(256);
  if (g.epi == MIA_EPI_STD) {
    if (g.out_f32) hipLaunchKernelGGL((gemm_nt_kernel<T, true, MIA_EPI_STD, STAGE>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_nt_kernel<T, false, MIA_EPI_STD, STAGE>), grid, block, 0, s, g);
  } else if (g.epi == MIA_EPI_QKV_VT) {
    hipLaunchKernelGGL((gemm_nt_kernel<T, false, MIA_EPI_QKV_VT, STAGE>), grid, block, 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_nt_kernel<T, false, MIA_EPI_HEADMAJOR, STAGE>), grid, block, 0, s, g);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename T>
int launch_t(const GemmArgs& g, hipStream_t s) {
  if (g.variant == 0) return launch_ts<T, 0>(g, s);
  if (g.variant == 1) return launch_ts<T, 1>(g, s);
  if (g.variant == 2) return launch_256<T>(g, s);
  // the 8-phase kernel addresses A and W through 32-bit byte offsets from the (per-batch) base and needs two K-tiles
  const bool ok8 = g.K >= 2 * BK && (int64_t)g.M * g.lda * 2 < (1ll << 32) && (int64_t)g.N * g.K * 2 < (1ll << 32);
  if (g.variant == 4) return ok8 ? launch_8ph<T>(g, s) : launch_256<T>(g, s);
  // auto: the 256^2 tile needs enough tiles to fill 256 CUs; small problems keep the 128^2 tile
  const long tiles256 = (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * (g.batch > 0 ? g.batch : 1);
  if (ok8 && (g.ln_gamma || g.ln_stat)) return launch_8ph<T>(g, s);      // (gemm_uses_8ph: hand-over pairs at every size)
  if (tiles256 < 256) return launch_ts<T, 1>(g, s);
  return ok8 ? launch_8ph<T>(g, s) : launch_256<T>(g, s);
}

}  // namespace

static bool gemm_uses_8ph(const GemmArgs& g) {
  const bool ok8 = g.K >= 2 * BK && (int64_t)g.M * g.lda * 2 < (1ll << 32) && (int64_t)g.N * g.K * 2 < (1ll << 32);
  if (!ok8) return false;
  if (g.variant == 4) return true;
  if (g.variant != 3) return false;
  // a LayerNorm hand-over pair runs on this kernel at EVERY size: its arithmetic (16-bit x * gamma operand, statistics applied in the
  // epilogue) differs from the LayerNorm kernel's in the last bit, and a clip's features must not depend on the batch it sits in
  if (g.ln_gamma || g.ln_stat) return true;
  return (long)((g.M + 255) / 256) * ((g.N + 255) / 256) * (g.batch > 0 ? g.batch : 1) >= 256;
}

bool mia_gemm_ln_ok(const GemmArgs& g) {
  if (!gemm_uses_8ph(g) || g.epi != MIA_EPI_STD || g.batch > 1) return false;
  if (g.ln_gamma) {          // producer: the fp32 row form of the epilogue (rows32)
    if (!g.out_f32 || (g.N & 63) || (g.ldc & 3) || ((uintptr_t)g.C & 15) || (g.strideC & 3) || !g.ln_out || !g.ln_part || (g.ln_ld & 3) ||
        ((uintptr_t)g.ln_out & 7) || ((uintptr_t)g.ln_gamma & 15)) return false;
    if (g.R && ((g.ldr & 3) || ((uintptr_t)g.R & 15) || (g.strideR & 3))) return false;
  }
  if (g.ln_stat) {           // consumer: the 16-bit row form (rows16)
    if (g.out_f32 || g.R || (g.N & 7) || (g.ldc & 7) || ((uintptr_t)g.C & 15) || (g.strideC & 7) || !g.ln_c1 || ((uintptr_t)g.ln_c1 & 15) || !g.bias) return false;
  }
  return true;
}

const char* mia_gemm_check(const GemmArgs& g) {
  if ((g.ln_gamma || g.ln_stat) && !mia_gemm_ln_ok(g)) return "gemm: the LayerNorm hand-over fields need the 8-phase kernel's row epilogues (mia_gemm_ln_ok)";
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return "gemm: M, N, K must be > 0";
  if (g.K % BK != 0) return "gemm: K must be a multiple of 64";
  if (g.lda % 8 != 0) return "gemm: lda must be a multiple of 8 elements (16-byte rows)";
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return "gemm: A and W must be 16-byte aligned";
  if (g.strideA % 8 != 0) return "gemm: strideA must be a multiple of 8 elements";
  if (g.epi != MIA_EPI_STD) {
    if (g.out_f32) return "gemm: special epilogues emit 16-bit output only";
    if (g.T <= 0 || g.H <= 0) return "gemm: special epilogue needs T and H";
    if (g.batch > 1) return "gemm: special epilogues use the flattened M = B*T form";
    if (g.epi == MIA_EPI_QKV_VT && (g.N != 3 * g.H * 64 || g.ldc % 4 != 0 || g.Tpad < g.T || !g.C2 || g.T % 4 != 0 || g.Tpad % 4 != 0)) return "gemm: bad QKV_VT arguments";
    if (g.epi == MIA_EPI_HEADMAJOR && g.N != g.H * 64) return "gemm: HEADMAJOR needs N == H*64";
  }
  return nullptr;
}

int mia_gemm_launch(const GemmArgs& g, int dtype, hipStream_t s) {
  return dtype == MIA_F16 ? launch_t<F16>(g, s) : launch_t<BF16>(g, s);
}
