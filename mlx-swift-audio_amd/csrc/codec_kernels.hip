// codec_kernels.hip -- fp32 kernels of the neural-codec decoders (SNAC, DAC) on gfx950.
//
// Activations are channels-last, time-major fp32 [T][C] throughout (the reference hops between [B,C,T] and [B,T,C] at
// every layer, SNACDecoder.swift:265-270, ResidualUnit.swift:62-78; here nothing is ever transposed).
//
//  * conv_gemm_f32   every dense convolution / transposed convolution / 1x1 projection as ONE tap-structured GEMM on the
//                    exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): A(m, tap, c) = pre(X[m + tap*dil - pad][c]) with an
//                    optional snake prologue x + sin^2(a x)/(a + 1e-9) applied while staging A into LDS; a transposed
//                    convolution with kernel 2*stride is `stride` such GEMMs (grid.z = output phase) over two-row windows,
//                    writing every `stride`-th output row -- no zero-stuffing, no scatter-add.
//                    Epilogue: bias, optional residual add, optional noise modulation r + noise[t]*acc, optional tanh.
//  * dwconv_snake    depthwise k-tap dilated conv with snake before and after (SNAC residual unit head), HBM-bound.
//  * conv_out1       snake -> conv k (C -> 1) -> tanh output stage.
//  * embed_codes     codebook gather + folded weight-normed 1x1 out_proj + stride expansion + level sum.
//  * noise_mod1      x[t][:] += noise[t] * (x[t][:] . w)   (SNAC NoiseBlock with one modulation channel).
#include "codec.h"
#include "mia_device.h"

namespace {

__device__ __forceinline__ float snake_f(float x, float a, float ra) {
  const float s = __sinf(a * x);
  return x + ra * (s * s);
}
// sin with an explicit two-constant Cody-Waite reduction in front of v_sin_f32 (which takes revolutions): k = rint(x / 2 pi),
// r = (x - k C1) - k C2 with C1 + C2 = 2 pi split so that k C1 is exact for |k| < 2^12 -- absolute error ~2e-7 for |x| < 2.5e4,
// six instructions instead of libm sinf's ~40 (its Payne-Hanek path is compiled in for every call).  The snake prologue runs it on
// every staged activation of every codec GEMM tile.
__device__ __forceinline__ float sin_cw(float x) {
  const float k = rintf(x * 0.15915494309189535f);
  float r = fmaf(-k, 6.28125f, x);                       // C1: 2 pi to 12 significant bits
  r = fmaf(-k, 1.9353071795864769e-03f, r);              // C2 = 2 pi - C1
  return __builtin_amdgcn_sinf(r * 0.15915494309189535f);
}
// variant used where the argument can be large (snake_f's bare v_sin_f32 loses accuracy with |a x|)
__device__ __forceinline__ float snake_p(float x, float a, float ra) {
  const float s = sin_cw(a * x);
  return x + ra * (s * s);
}


// WM / WN = 32-row / 32-column MFMA tiles per wave (4 waves as 2 x 2): block tile (64 WM) x (64 WN).  <2,2> = 128 x 128 for large
// problems, <2,1> = 128 x 64 for narrow outputs, <1,1> = 64 x 64 when 128-row tiles would leave most of the 256 CUs idle.
// KC = 32-wide K chunks per pipeline stage: the next stage's global loads are issued before the MFMAs of the current one, so a stage
// must hold more MFMA time than one HBM / L2 round trip -- with 64 x 64 tiles a 32-wide stage is only 0.4 us of MFMAs (measured:
// 3.5x slower than the MFMA bound), hence KC = 4 there.  A chunk never straddles a tap (Cin % 32 == 0); chunks past K are zero.
template <int WM, int WN, int KC>
__global__ __launch_bounds__(256) void conv_gemm_f32(ConvGemmArgs g) {
  constexpr int GBM = 64 * WM;
  constexpr int GBN = 64 * WN;
  // LDS rows hold the K-tile PERMUTED: even k in the first half, odd k in the second.  v_mfma_f32_32x32x2_f32 wants k = 2 s + (lane >> 5)
  // for step s, so a lane's operands of 4 consecutive steps are 4 consecutive floats of "its" half: one ds_read_b128 per 4 MFMA steps
  // instead of four ds_read_b32 (the kernel's rate tracked the number of LDS operand reads per MFMA, not the MFMA count).
  constexpr int GBK = 32 * KC, GH = GBK / 2, GSTR = GBK + 4;   // row stride 16-byte aligned; 36 floats: 8 consecutive rows hit distinct 16-B slots
  __shared__ float As[GBM * GSTR];
  __shared__ float Bs[GBN * GSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
  int z = blockIdx.z, u = 0;
  if (g.n_seq > 1) { const int pps = (int)gridDim.z / g.n_seq; u = z / pps; z -= u * pps; }     // grid.z = sequence x phase
  const float* __restrict__ W = g.W + (int64_t)z * g.w_phase_stride;
  const float* __restrict__ X = g.X + ((int64_t)z * g.x_phase_step + (int64_t)u * g.x_seq_step) * g.ldx;   // stacked sequences read their own rows only
  const int T_valid = g.seq_len ? min(g.T_in, g.seq_len[u]) : g.phase_len ? min(g.T_in, g.phase_len[z]) : g.T_in;   // padded sequence: rows past its own length are zeros
  // the sequence's rows of Y / R / R2 as (wave-uniform) base pointers: the epilogue's row arithmetic stays what it was -- added to the row
  // index per element it cost 4 VGPRs, and the 128 x 128 tile went from 168 to 172: 3 -> 2 waves per SIMD (SNAC + 10 %, DAC + 15 %)
  const int64_t y_seq = (int64_t)u * g.y_seq_step;
  float* __restrict__ Yb = g.Y + y_seq * g.ldy;
  const float* __restrict__ Rb = g.R ? g.R + y_seq * g.ldr : nullptr;
  const float* __restrict__ R2b = g.R2 ? g.R2 + y_seq * g.ldr : nullptr;
  const int Ktot = g.taps * g.Cin;
  const int64_t ldw = g.ldw > 0 ? g.ldw : Ktot;

  // staging coordinates: per chunk, A: 2 WM rows x float4 per thread, B: 2 WN rows x float4 per thread.
  // Loads are UNCONDITIONAL (row / column indices clamped into the tensors) and the out-of-range zeroing plus the snake / leaky-ReLU
  // prologue are applied when the registers are written to LDS: a predicated load (`if (ok) v = load`) makes the compiler drain
  // vmcnt between consecutive loads, which serialised the 4-8 loads of a stage on their full memory latency.
  const int s_row = tid >> 3, s_col = (tid & 7) * 4;
  float4 ra[KC][2 * WM], rb[KC][2 * WN];
  uint32_t amask = 0, bmask = 0;            // bit (ch * 8 + i): element is inside the tensor
  // Per-thread row bases, once per tile.  The stage loop then only adds a wave-uniform tap offset and clamps the OFFSET (a masked
  // element's address just has to stay inside the tensor): no integer division (k0 / Cin) and no 64-bit multiply per row and chunk --
  // they were ~100 quarter-rate integer multiplies per K stage, in a loop whose wave also does the MFMAs.
  int xr0[2 * WM], xo0[2 * WM];
  int64_t wb[2 * WN];
  uint32_t mok = 0, nok = 0;
  const int ldx_i = (int)g.ldx, xo_max = (g.T_in - 1) * ldx_i;       // host-checked: every row offset fits 31 bits
#pragma unroll
  for (int i = 0; i < 2 * WM; ++i) {
    const int m = m0 + s_row + 32 * i;
    xr0[i] = m * g.x_row_mul - g.pad;
    xo0[i] = xr0[i] * ldx_i;
    mok |= (m < g.M ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < 2 * WN; ++i) {
    const int n = n0 + s_row + 32 * i;
    wb[i] = (int64_t)(n < g.N ? n : g.N - 1) * ldw + s_col;
    nok |= (n < g.N ? 1u : 0u) << i;
  }
  int cc_s[KC];                             // per chunk of the staged stage: channel offset inside its tap
  int tap_b = 0, cb = 0;                    // the stage's first chunk starts at k = tap_b * Cin + cb (stages are visited in order)
  auto load_tiles = [&](int kbase) {
    if (kbase > 0) {                        // advance by one stage (wave-uniform scalar work)
      cb += GBK;
      while (cb >= g.Cin) { cb -= g.Cin; ++tap_b; }
    }
    amask = 0; bmask = 0;
#pragma unroll
    for (int ch = 0; ch < KC; ++ch) {
      int tap = tap_b, cc = cb + 32 * ch;
#pragma unroll
      for (int w2 = 0; w2 < KC; ++w2) if (cc >= g.Cin) { cc -= g.Cin; ++tap; }       // Cin >= 32: at most KC - 1 wraps inside a stage
      const bool kin = tap < g.taps;        // == (kbase + 32 ch < Ktot)
      if (!kin) { tap = g.taps - 1; cc = g.Cin - 32; }
      cc_s[ch] = cc;
      const int k0 = tap * g.Cin + cc;
      const int tapd = tap * g.dil, tapo = tapd * ldx_i;
#pragma unroll
      for (int i = 0; i < 2 * WM; ++i) {
        const int xr = xr0[i] + tapd;
        const bool ok = kin && ((mok >> i) & 1u) && xr >= 0 && xr < T_valid;
        const int xo = min(max(xo0[i] + tapo, 0), xo_max);
        ra[ch][i] = *reinterpret_cast<const float4*>(X + xo + cc + s_col);
        amask |= (ok ? 1u : 0u) << (ch * 8 + i);
      }
#pragma unroll
      for (int i = 0; i < 2 * WN; ++i) {
        rb[ch][i] = *reinterpret_cast<const float4*>(W + wb[i] + k0);
        bmask |= ((kin && ((nok >> i) & 1u)) ? 1u : 0u) << (ch * 8 + i);
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int ch = 0; ch < KC; ++ch) {
      float al[4] = {0.f, 0.f, 0.f, 0.f}, ral[4] = {0.f, 0.f, 0.f, 0.f};
      if (g.alpha) {
        const int c0 = cc_s[ch] + s_col;                  // channel offset of this chunk inside its tap (set by load_tiles)
#pragma unroll
        for (int j = 0; j < 4; ++j) { al[j] = g.alpha[c0 + j]; ral[j] = g.ralpha ? g.ralpha[c0 + j] : 1.0f / (al[j] + 1e-9f); }
      }
#pragma unroll
      for (int i = 0; i < 2 * WM; ++i) {
        float4 v = ra[ch][i];
        if (g.alpha) {
          v.x = snake_p(v.x, al[0], ral[0]); v.y = snake_p(v.y, al[1], ral[1]);
          v.z = snake_p(v.z, al[2], ral[2]); v.w = snake_p(v.w, al[3], ral[3]);
        } else if (g.lrelu_slope > 0.f) {
          v.x = v.x > 0.f ? v.x : v.x * g.lrelu_slope; v.y = v.y > 0.f ? v.y : v.y * g.lrelu_slope;
          v.z = v.z > 0.f ? v.z : v.z * g.lrelu_slope; v.w = v.w > 0.f ? v.w : v.w * g.lrelu_slope;
        }
        if (!((amask >> (ch * 8 + i)) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = As + (s_row + 32 * i) * GSTR + (32 * ch + s_col) / 2;
        *reinterpret_cast<float2*>(d) = make_float2(v.x, v.z);            // even k
        *reinterpret_cast<float2*>(d + GH) = make_float2(v.y, v.w);       // odd k
      }
#pragma unroll
      for (int i = 0; i < 2 * WN; ++i) {
        float4 v = rb[ch][i];
        if (!((bmask >> (ch * 8 + i)) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = Bs + (s_row + 32 * i) * GSTR + (32 * ch + s_col) / 2;
        *reinterpret_cast<float2*>(d) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(d + GH) = make_float2(v.y, v.w);
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int a_off = (wr * 32 * WM + (lane & 31)) * GSTR + (lane >> 5) * GH;
  const int b_off = (wc * 32 * WN + (lane & 31)) * GSTR + (lane >> 5) * GH;
  const int nk = (Ktot + GBK - 1) / GBK;
  load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                       // previous tile fully consumed
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles((kt + 1) * GBK);   // global loads fly under the MFMAs below
    // LDS operand reads are software-pipelined by hand: batch i+1 (PB K-steps) is read before the MFMAs of batch i issue.  Left to
    // the compiler every MFMA waited on the ds_read issued just before it (~120 cycles of LDS latency per 64-cycle MFMA: measured
    // 2.1x the MFMA bound per K-tile).
    constexpr int NS = GBK / 2, PB = 4;
    f32x4 fa[2][WM], fb[2][WN];                  // one 128-bit read = the operands of PB = 4 consecutive MFMA steps
    auto lds_batch = [&](int buf, int s0) {
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[buf][i] = *reinterpret_cast<const f32x4*>(As + a_off + i * 32 * GSTR + s0);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[buf][j] = *reinterpret_cast<const f32x4*>(Bs + b_off + j * 32 * GSTR + s0);
    };
    lds_batch(0, 0);
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += PB) {
      const int cur = (s0 / PB) & 1;
      if (s0 + PB < NS) lds_batch(cur ^ 1, s0 + PB);
      __builtin_amdgcn_sched_barrier(0);       // keep the reads of the next batch ahead of this batch's MFMAs
#pragma unroll
      for (int u = 0; u < PB; ++u)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][u], fb[cur][j][u], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue: lane owns column n = ..+(lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + wc * 32 * WN + j * 32 + (lane & 31);
    if (n >= g.N) continue;
    const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= g.M) continue;
        const int64_t yr = (int64_t)m * g.y_row_mul + g.y_row_off + (int64_t)z * g.y_phase_step;
        if (yr < 0 || yr >= g.T_out) continue;
        float v = acc[i][j][r] + bias;
        if (g.gelu == 1) v = gelu_erf(v);
        else if (g.gelu == 2) v = v > 0.f ? v : (__expf(v) - 1.0f);
        else if (g.gelu == 3) v = fabsf(v);
        else if (g.gelu == 4) v = v / (1.0f + __expf(-v));
        else if (g.gelu == 5) v = v > 0.f ? v : 0.01f * v;
        else if (g.gelu == 6) v = fmaxf(v, 0.f);
        if (g.noise) v = Rb[yr * g.ldr + n] + g.noise[yr] * v;
        else if (Rb) v += Rb[yr * g.ldr + n];
        if (g.out_scale != 0.f) v *= g.out_scale;
        if (R2b) v += R2b[yr * g.ldr + n];
        if (g.tanh_out) v = tanhf(v);
        Yb[yr * g.ldy + n] = v;
      }
  }
}

// y[t][c] = post( bias[c] + sum_k w[k][c] * pre(x[t + (k - K/2)*dil][c]) ),  pre/post = snake with alpha_pre/alpha_post (optional)
__global__ __launch_bounds__(256) void dwconv_snake(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ a_pre,
                                                    const float* __restrict__ a_post, int T, int C, int K, int dil) {
  const int c4n = C >> 2;
  const int64_t total = (int64_t)T * c4n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e / c4n), c = (int)(e - (int64_t)t * c4n) * 4;
    float ap[4] = {0, 0, 0, 0}, rap[4] = {0, 0, 0, 0};
    if (a_pre) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ap[j] = a_pre[c + j]; rap[j] = 1.0f / (ap[j] + 1e-9f); }
    }
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bias ? bias[c + j] : 0.f;
    for (int k = 0; k < K; ++k) {
      const int ts = t + (k - K / 2) * dil;
      if (ts < 0 || ts >= T) continue;
      const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)ts * C + c);
      const float4 wk = *reinterpret_cast<const float4*>(w + (int64_t)k * C + c);
      float xv[4] = {v.x, v.y, v.z, v.w};
      const float wv[4] = {wk.x, wk.y, wk.z, wk.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (a_pre) xv[j] = snake_p(xv[j], ap[j], rap[j]);
        acc[j] = fmaf(wv[j], xv[j], acc[j]);
      }
    }
    if (a_post) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float a = a_post[c + j]; acc[j] = snake_p(acc[j], a, 1.0f / (a + 1e-9f)); }
    }
    *reinterpret_cast<float4*>(y + (int64_t)t * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// out[t] = tanh( bias + sum_{k,c} w[k*C + c] * snake(x[t + k - K/2][c]) )      (C -> 1 output convolution)
__global__ __launch_bounds__(256) void conv_out1(const float* __restrict__ x, float* __restrict__ out, const float* __restrict__ w,
                                                 const float* __restrict__ bias, const float* __restrict__ alpha, int T, int C, int K) {
  extern __shared__ float sm[];        // w[K*C] | alpha[C] | ralpha[C]
  float* sw = sm;
  float* sa = sm + K * C;
  float* sr = sa + C;
  for (int i = threadIdx.x; i < K * C; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < C; i += 256) { const float a = alpha ? alpha[i] : 0.f; sa[i] = a; sr[i] = 1.0f / (a + 1e-9f); }
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float acc = bias ? bias[0] : 0.f;
  for (int k = 0; k < K; ++k) {
    const int ts = t + k - K / 2;
    if (ts < 0 || ts >= T) continue;
    const float* xr = x + (int64_t)ts * C;
    for (int c = 0; c < C; c += 4) {
      const float4 v = *reinterpret_cast<const float4*>(xr + c);
      float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s = alpha ? snake_p(xv[j], sa[c + j], sr[c + j]) : xv[j];
        acc = fmaf(sw[k * C + c + j], s, acc);
      }
    }
  }
  out[t] = tanhf(acc);
}

// z[t][ch] = sum_levels ( b_l[ch] + sum_j cb_l[code_l[t / stride_l]][j] * Weff_l[ch][j] )
__global__ __launch_bounds__(256) void embed_codes(EmbedArgs a, float* __restrict__ z, int T, int C) {
  const int64_t total = (int64_t)T * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e / C), ch = (int)(e - (int64_t)t * C);
    float acc = 0.f;
    for (int l = 0; l < a.n_levels; ++l) {
      if (!a.codes[l]) continue;
      const int code = a.codes[l][t / a.stride[l]];
      const float* cb = a.codebook[l] + (int64_t)code * a.cb_dim;
      const float* wv = a.weff[l] + (int64_t)ch * a.cb_dim;
      float s = 0.f;
      for (int j = 0; j < a.cb_dim; ++j) s = fmaf(cb[j], wv[j], s);
      acc += s + a.bias[l][ch];
    }
    z[e] = acc;
  }
}

// x[t][:] += noise[t] * dot(x[t][:], w)      one wave per row
__global__ __launch_bounds__(256) void noise_mod1(float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ noise, int T, int C) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T) return;
  float* xr = x + (int64_t)t * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(xr[c], w[c], s);
  s = wave_sum(s) * noise[t];
  for (int c = lane; c < C; c += 64) xr[c] += s;
}

// ---- DAC encoder front: y[t][c] = b[c] + sum_k w[c][k] x[t + k - pad]   (Cin = 1 -> C channels; HBM-bound: one write of T x C floats)
__global__ __launch_bounds__(256) void conv_in1(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                                                int64_t T, int C, int K, int pad) {
  const int64_t total = T * (C / 4);
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t t = e / (C / 4);
    const int c = (int)(e - t * (C / 4)) * 4;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const int64_t xi = t + k - pad;
        acc += w[(c + j) * K + k] * ((xi >= 0 && xi < T) ? x[xi] : 0.f);
      }
      o[j] = acc + bias[c + j];
    }
    *reinterpret_cast<float4*>(y + t * C + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- one residual-VQ stage per time step (DACQuantize.swift:87-115,147-190): nearest codebook entry of the L2-normalised in_proj output
// (first index wins ties, like argMax(-dist)), then residual[t][:] -= out_proj(zE + (codebook[idx] - zE)).
struct VqArgs {
  const float* zE;       // [T][cd]   in_proj output of this stage
  const float* cbn;      // [cs][cd]  L2-normalised codebook (host, same formula as l2Normalize)
  const float* cbn_sq;   // [cs]      sum of squares of the normalised rows
  const float* cb;       // [cs][cd]  raw codebook
  const float* weff;     // [C][cd]   folded out_proj
  const float* bias;     // [C]
  float* residual;       // [T][C]
  int32_t* codes;        // [T]
  int T, C, cs, cd;
};
__global__ __launch_bounds__(256) void vq_assign(VqArgs a) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ float s_e[16], s_en[16], s_z[16];
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < a.cd) s_e[tid] = a.zE[(int64_t)t * a.cd + tid];
  __syncthreads();
  if (tid == 0) {     // l2Normalize: x / max(sqrt(sum |x|^2), 1e-12)
    float ss = 0.f;
    for (int d = 0; d < a.cd; ++d) ss += fabsf(s_e[d]) * fabsf(s_e[d]);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float sq = 0.f;
    for (int d = 0; d < a.cd; ++d) { s_en[d] = s_e[d] / nrm; sq += s_en[d] * s_en[d]; }
    s_en[a.cd] = sq;
  }
  __syncthreads();
  const float se = s_en[a.cd];
  float best = INFINITY; int bi = 0x7fffffff;
  for (int j = tid; j < a.cs; j += 256) {
    float dot = 0.f;
    for (int d = 0; d < a.cd; ++d) dot += s_en[d] * a.cbn[(int64_t)j * a.cd + d];
    const float dist = (se - 2.0f * dot) + a.cbn_sq[j];
    if (dist < best) { best = dist; bi = j; }         // ascending j per thread: the first minimum is kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wave] = best; s_i[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w2 = 1; w2 < 4; ++w2) if (s_v[w2] < best || (s_v[w2] == best && s_i[w2] < bi)) { best = s_v[w2]; bi = s_i[w2]; }
    a.codes[t] = bi;
    for (int d = 0; d < a.cd; ++d) s_z[d] = s_e[d] + (a.cb[(int64_t)bi * a.cd + d] - s_e[d]);   // straight-through form (:62)
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += 256) {
    float acc = 0.f;
    for (int d = 0; d < a.cd; ++d) acc += s_z[d] * a.weff[(int64_t)c * a.cd + d];
    a.residual[(int64_t)t * a.C + c] -= acc + a.bias[c];
  }
}

}  // namespace

const char* codec_conv_gemm_check(const ConvGemmArgs& g) {
  if (g.M <= 0 || g.N <= 0 || g.Cin <= 0 || g.taps <= 0) return "conv_gemm: bad shape";
  if (g.Cin % 32) return "conv_gemm: Cin must be a multiple of 32";
  if (g.ldx % 4 || ((uintptr_t)g.X & 15) || ((uintptr_t)g.W & 15)) return "conv_gemm: X rows / W must be 16-byte aligned";
  if (g.noise && !g.R) return "conv_gemm: noise modulation needs the residual input";
  if (g.n_seq < 1 || (g.n_seq > 1 && (g.x_seq_step < g.T_in || g.y_seq_step <= 0 || g.noise || g.x_phase_step))) return "conv_gemm: bad stacked-sequence arguments";
  if ((int64_t)g.M * g.x_row_mul + (int64_t)g.taps * g.dil > 0x7fffffffLL) return "conv_gemm: row index exceeds 31 bits";
  if (((int64_t)g.M * g.x_row_mul + (int64_t)g.taps * g.dil + g.pad + g.T_in) * g.ldx > 0x7fffffffLL) return "conv_gemm: row offset exceeds 31 bits";
  return nullptr;
}

// ---- algorithmic-byte accounting (profiling aid behind mia_profile_codec_bytes; SURVEY.md 8(d): "SNAC / DAC / HiFT decode: HBM-bound,
// report bytes as the sum over fused ops of one read + one write of the activation tensor").  Every launcher below adds what its launch
// must move at minimum -- each distinct input row once, each output element once, the weights once, residual inputs once -- to a
// counter of the calling host thread.
static thread_local double t_alg_bytes = 0.0;
double codec_alg_bytes(bool reset) { const double v = t_alg_bytes; if (reset) t_alg_bytes = 0.0; return v; }

static void account_conv_gemm(const ConvGemmArgs& g, int phases) {
  if (g.n_seq > 1) {                                                  // n_seq independent problems of the n_seq = 1 shape
    ConvGemmArgs one = g; one.n_seq = 1;
    const double before = t_alg_bytes;
    account_conv_gemm(one, phases);
    t_alg_bytes = before + (t_alg_bytes - before) * g.n_seq;
    return;
  }
  const bool stacked = g.x_phase_step != 0;                          // grid.z = stacked sequences (own X rows); else = stride phases of ONE input
  const double rows_in = std::min<double>((double)g.T_in, (double)g.M * g.x_row_mul + (double)g.taps * g.dil);
  double b = rows_in * g.Cin * 4.0 * (stacked ? phases : 1);
  b += (double)g.N * g.taps * g.Cin * 4.0 * (g.w_phase_stride ? phases : 1);
  const double out = (double)g.M * g.N * 4.0 * phases;
  b += out;
  if (g.R) b += out;
  if (g.R2) b += out;
  if (g.noise) b += (double)g.M * 4.0 * phases;
  t_alg_bytes += b;
}

int codec_conv_gemm_launch(const ConvGemmArgs& g, int phases_per_seq, hipStream_t s) {
  account_conv_gemm(g, phases_per_seq);
  const int phases = phases_per_seq * (g.n_seq > 1 ? g.n_seq : 1);
  // 128-row tiles only when they still give every CU work; otherwise 64 x 64 tiles (4x the workgroups)
  const int64_t big_blocks = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * phases;
  if (big_blocks < 384) {
    dim3 grid((g.M + 63) / 64, (g.N + 63) / 64, phases);
    // at most ~1 workgroup per CU: nothing else hides the global-load latency of a stage, so stages are 4 chunks deep
    if ((int64_t)grid.x * grid.y * grid.z <= 320 && g.taps * g.Cin >= 256) hipLaunchKernelGGL((conv_gemm_f32<1, 1, 4>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((conv_gemm_f32<1, 1, 1>), grid, dim3(256), 0, s, g);
  } else if (g.N > 64) {
    dim3 grid((g.M + 127) / 128, (g.N + 127) / 128, phases);
    hipLaunchKernelGGL((conv_gemm_f32<2, 2, 1>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid((g.M + 127) / 128, (g.N + 63) / 64, phases);
    hipLaunchKernelGGL((conv_gemm_f32<2, 1, 1>), grid, dim3(256), 0, s, g);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int codec_dwconv_launch(const float* x, float* y, const float* w, const float* bias, const float* a_pre, const float* a_post, int T, int C,
                        int K, int dil, hipStream_t s) {
  if (C % 4) return -1;
  t_alg_bytes += 2.0 * T * C * 4.0 + (double)K * C * 4.0;
  const int64_t total = (int64_t)T * (C / 4);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(dwconv_snake, dim3(grid), dim3(256), 0, s, x, y, w, bias, a_pre, a_post, T, C, K, dil);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int codec_conv_out1_launch(const float* x, float* out, const float* w, const float* bias, const float* alpha, int T, int C, int K, hipStream_t s) {
  if (C % 4) return -1;
  t_alg_bytes += (double)T * C * 4.0 + (double)T * 4.0;
  const size_t lds = (size_t)(K * C + 2 * C) * 4;
  hipLaunchKernelGGL(conv_out1, dim3((T + 255) / 256), dim3(256), lds, s, x, out, w, bias, alpha, T, C, K);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int codec_embed_launch(const EmbedArgs& a, float* z, int T, int C, hipStream_t s) {
  const int64_t total = (int64_t)T * C;
  t_alg_bytes += (double)total * 4.0;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(embed_codes, dim3(grid), dim3(256), 0, s, a, z, T, C);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int codec_noise1_launch(float* x, const float* w, const float* noise, int T, int C, hipStream_t s) {
  t_alg_bytes += 2.0 * T * C * 4.0 + (double)T * 4.0;
  hipLaunchKernelGGL(noise_mod1, dim3((T + 3) / 4), dim3(256), 0, s, x, w, noise, T, C);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int codec_conv_in1_launch(const float* x, float* y, const float* w, const float* bias, int64_t T, int C, int K, int pad, hipStream_t s) {
  if (C % 4) return -1;
  const int64_t total = T * (C / 4);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(conv_in1, dim3(grid), dim3(256), 0, s, x, y, w, bias, T, C, K, pad);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int codec_vq_assign_launch(const float* zE, const float* cbn, const float* cbn_sq, const float* cb, const float* weff, const float* bias,
                           float* residual, int32_t* codes, int T, int C, int cs, int cd, hipStream_t s) {
  if (cd > 15 || T <= 0) return -1;
  VqArgs a{zE, cbn, cbn_sq, cb, weff, bias, residual, codes, T, C, cs, cd};
  hipLaunchKernelGGL(vq_assign, dim3(T), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
