// hift.hip -- HiFT-Net vocoder of CosyVoice2 (80-bin mel @ 50 Hz -> 24 kHz waveform), fp32, gfx950.
//
// Replaces CosyHiFTGenerator (TTS/CosyVoice2/HiFiGAN/CosyHiFTGenerator.swift:261-512): CosyF0Predictor (:212-256),
// SourceModuleHnNSF2 / SineGen2 (:66-204), decode (:412-475) with HiFiGANResBlock / Snake / stftHiFiGAN / istftHiFiGAN
// (Codec/S3Gen/HiFiGAN.swift:30-130,257-367).
//
// Layout: every activation is time-major fp32 [T][C] (the reference swaps axes around every convolution).  Every convolution --
// plain, dilated, strided (source_downs) and transposed (ups) -- runs on the exact-fp32 MFMA tap GEMM of codec_kernels.hip:
//   * Snake (alpha clamped to |a| >= 1e-4, HiFiGAN.swift:53-66) and leaky-ReLU are prologues applied while A tiles are staged;
//   * ELU / abs / residual add / "mean of three ResBlocks" (scale 1/3 + accumulate) / "h + source" are epilogues,
// so a ResBlock is 2 launches per dilation and no elementwise pass exists anywhere in the decoder;
//   * ConvTransposed1d with kernel K and stride s is s phase-GEMMs over ceil(K/s)-row windows (taps that fall outside the kernel
//     carry zero weights) -- no zero-stuffing, no scatter;
//   * the iSTFT is a per-frame 16-point synthesis followed by a GATHER overlap-add (each output sample sums its <= 4 frames in a
//     fixed order): deterministic, no atomics (the reference scatter-adds with at[].add, HiFiGAN.swift:340-357).
// The sine source replays the reference's float32 operation order with unfused multiplies/adds (its phases reach ~1e5 rad, where
// a different association changes the waveform visibly); the Gaussian it draws is an explicit input.
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "codec.h"
#include "mia_device.h"
#include "mia_internal.h"

namespace {

struct Conv {
  float* w = nullptr; float* b = nullptr; float* alpha = nullptr; float* ralpha = nullptr;
  int N = 0, Cin = 0, taps = 1, dil = 1, pad = 0, stride = 1;
  int K = 0;   // transposed conv: kernel size (taps = ceil(K / stride))
};
struct ResBlock { Conv c1[4], c2[4]; };

__constant__ float c_cos16[16] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f, -0.38268343236508977f,
                                  -0.70710678118654752f, -0.92387953251128674f, -1.f, -0.92387953251128674f, -0.70710678118654752f,
                                  -0.38268343236508977f, 0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__constant__ float c_sin16[16] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.f, 0.92387953251128674f,
                                  0.70710678118654752f, 0.38268343236508977f, 0.f, -0.38268343236508977f, -0.70710678118654752f,
                                  -0.92387953251128674f, -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
// periodic Hann, n_fft 16: 0.5 (1 - cos(2 pi n / 16))  (HiFiGAN.swift:15-20)
__device__ __forceinline__ float hann16(int n) { return 0.5f * (1.0f - c_cos16[n]); }

// mel [C][T] (reference layout) -> [T][Cp], channels C..Cp-1 zero
__global__ __launch_bounds__(256) void hift_pack_mel(const float* __restrict__ mel, float* __restrict__ out, int C, int Cp, int T) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < T) ? mel[(int64_t)c * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < Cp) out[(int64_t)t * Cp + c] = tile[tx][i];
  }
}

// P[i][h] = ((cumsum_i(rad_h) * 2) * pi) * up,  rad_h[i] = ((f0[i] (h+1)) / sr) mod 1     (SineGen2.f02sine, :96-128: the 480:1
// linear downsample of the piecewise-constant full-rate track returns the frame value itself: 0.5 x + 0.5 x)
__global__ __launch_bounds__(256) void hift_phase_scan(const float* __restrict__ f0, float* __restrict__ P, int T, int H, float sr,
                                                       float up) {
  __shared__ float sf[1024];
  float c = 0.f;
  const int h = threadIdx.x;
  for (int base = 0; base < T; base += 1024) {
    const int n = min(1024, T - base);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) sf[i] = f0[base + i];
    __syncthreads();
    if (h < H) {
      const float hm = (float)(h + 1);
      for (int i = 0; i < n; ++i) {
        const float rad = fmodf(__fdiv_rn(__fmul_rn(sf[i], hm), sr), 1.0f);
        c = __fadd_rn(c, rad);
        P[(int64_t)(base + i) * H + h] = __fmul_rn(__fmul_rn(__fmul_rn(c, 2.0f), 3.14159274101257324f), up);
      }
    }
  }
}

// one thread per output sample: linear up-interpolation of P (align_corners = False rule of linearInterpolate1d, :17-60), sin,
// voiced mask, additive noise, tanh(Linear_{H -> 1})   (:134-153, :190-203); the first cache_len samples come from `cache`.
__global__ __launch_bounds__(256) void hift_source_kernel(const float* __restrict__ f0, const float* __restrict__ P,
                                                          const float* __restrict__ noise, const float* __restrict__ lw, float lb,
                                                          const float* __restrict__ cache, int cache_len, float* __restrict__ s, int T,
                                                          int up, int H, float ratio, float clip_hi, float amp, float sigma, float thr) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t L = (int64_t)T * up;
  if (j >= L) return;
  if (j < cache_len) { s[j] = cache[j]; return; }
  float idx = __fsub_rn(__fmul_rn(__fadd_rn((float)j, 0.5f), ratio), 0.5f);
  idx = fminf(fmaxf(idx, 0.f), clip_hi);
  const float lo = floorf(idx);
  const int ilo = (int)lo, ihi = min(ilo + 1, T - 1);
  const float wh = __fsub_rn(idx, lo), wl = __fsub_rn(1.0f, wh);
  const float f = f0[j / up];
  const float uv = f > thr ? 1.f : 0.f;
  const float namp = __fadd_rn(__fmul_rn(uv, sigma), __fdiv_rn(__fmul_rn(__fsub_rn(1.f, uv), amp), 3.0f));
  float acc = lb;
  for (int h = 0; h < H; ++h) {
    const float ph = __fadd_rn(__fmul_rn(P[(int64_t)ilo * H + h], wl), __fmul_rn(P[(int64_t)ihi * H + h], wh));
    float v = __fmul_rn(__fmul_rn(sinf(ph), amp), uv);
    if (noise) v = __fadd_rn(v, __fmul_rn(namp, noise[j * H + h]));
    acc = fmaf(lw[h], v, acc);
  }
  s[j] = tanhf(acc);
}

// stftHiFiGAN (HiFiGAN.swift:257-295): reflect pad 8, frames of 16 @ hop 4, periodic Hann, bins 0..8 -> out[f][0..8] = real,
// out[f][9..17] = imag (forward transform, e^{-i...}), out[f][18..31] = 0 (channel padding of the tap GEMM)
__global__ __launch_bounds__(256) void hift_stft(const float* __restrict__ s, float* __restrict__ out, int64_t L, int64_t F) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  float x[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    int64_t q = 4 * f + n - 8;
    q = q < 0 ? -q : (q >= L ? 2 * (L - 1) - q : q);
    x[n] = s[q] * hann16(n);
  }
  float4* o = reinterpret_cast<float4*>(out + f * 32);
  float r[32];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float re = 0.f, im = 0.f;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      re = fmaf(x[n], c_cos16[(k * n) & 15], re);
      im = fmaf(x[n], -c_sin16[(k * n) & 15], im);
    }
    r[k] = re; r[9 + k] = im;
  }
#pragma unroll
  for (int k = 18; k < 32; ++k) r[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
}

// conv_post row [18] -> windowed time frame [16]: mag = min(exp(h[:9]), 100), phase = sin(h[9:]); conjugate-symmetric 16-point
// inverse transform (real part), 1/16 normalisation, x Hann   (CosyHiFTGenerator.swift:452-455, HiFiGAN.swift:298-330)
__global__ __launch_bounds__(256) void hift_istft_frames(const float* __restrict__ hp, float* __restrict__ fr, int64_t F) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  float re[9], im[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float mag = fminf(expf(hp[f * 18 + k]), 100.0f);
    const float ph = sinf(hp[f * 18 + 9 + k]);
    re[k] = mag * cosf(ph); im[k] = mag * sinf(ph);
  }
  float4* o = reinterpret_cast<float4*>(fr + f * 16);
  float y[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    float a = re[0] + ((n & 1) ? -re[8] : re[8]);
#pragma unroll
    for (int k = 1; k < 8; ++k) a += 2.0f * (re[k] * c_cos16[(k * n) & 15] - im[k] * c_sin16[(k * n) & 15]);
    y[n] = a * 0.0625f * hann16(n);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = make_float4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]);
}

// gather overlap-add + window-square normalisation + trim 8 + clip   (HiFiGAN.swift:332-366, CosyHiFTGenerator.swift:466)
__global__ __launch_bounds__(256) void hift_overlap_add(const float* __restrict__ fr, float* __restrict__ pcm, int64_t F, int64_t L,
                                                        float limit) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  const int64_t j = t + 8;
  int64_t f_hi = j / 4; f_hi = f_hi < F - 1 ? f_hi : F - 1;
  int64_t f_lo = j >= 15 ? (j - 15 + 3) / 4 : 0;
  float acc = 0.f, ws = 0.f;
  for (int64_t f = f_lo; f <= f_hi; ++f) {
    const int n = (int)(j - 4 * f);
    const float w = hann16(n);
    acc += fr[f * 16 + n];
    ws = fmaf(w, w, ws);
  }
  const float v = acc / fmaxf(ws, 1e-8f);
  pcm[t] = fminf(fmaxf(v, -limit), limit);
}

}  // namespace

struct mia_hift {
  mia_ctx* ctx = nullptr;
  mia_hift_config cfg{};
  std::vector<void*> allocs;
  int Cp_mel = 0, H = 0, up = 0;
  Conv f0c[5], cls, pre, post, ups[4], sdown[4];
  ResBlock srb[4], rb[4][4];
  float* lw = nullptr; float lb = 0.f;
  // scratch arena (grow-only), carved per call
  float* arena = nullptr; size_t arena_floats = 0;
};

namespace {

struct HLoader {
  mia_hift* c;
  std::map<std::string, const mia_tensor_view*> by_name;
  std::string err;

  bool f32(const std::string& n, std::vector<float>& out, std::initializer_list<int64_t> shp) {
    auto it = by_name.find(n);
    if (it == by_name.end()) { if (err.empty()) err = "missing tensor '" + n + "'"; return false; }
    const mia_tensor_view* t = it->second;
    if (t->dtype != MIA_F32) { if (err.empty()) err = "tensor '" + n + "' must be float32"; return false; }
    int64_t numel = 1; bool ok = t->ndim == (int)shp.size(); int i = 0;
    for (int64_t s : shp) { if (ok && t->shape[i] != s) ok = false; ++i; }
    for (int k = 0; k < t->ndim; ++k) numel *= t->shape[k];
    if (!ok) { if (err.empty()) err = "tensor '" + n + "' has an unexpected shape"; return false; }
    out.assign((const float*)t->data, (const float*)t->data + numel);
    return true;
  }
  float* up(const std::vector<float>& v) {
    void* p = nullptr;
    if (hipMalloc(&p, v.size() * 4 + 64) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    c->allocs.push_back(p);
    if (hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    return (float*)p;
  }
  // Conv1d weight [Cout][K][Cin] -> tap-GEMM weights [Cout][K][Cp] (input channels zero-padded to a multiple of 32)
  bool conv(const std::string& p, int Cout, int K, int Cin, int dil, int pad, int stride, Conv& o) {
    std::vector<float> w, b;
    if (!f32(p + ".weight", w, {Cout, K, Cin}) || !f32(p + ".bias", b, {Cout})) return false;
    const int Cp = (Cin + 31) / 32 * 32;
    if (Cp != Cin) {
      std::vector<float> wp((size_t)Cout * K * Cp, 0.f);
      for (int co = 0; co < Cout; ++co) for (int k = 0; k < K; ++k) for (int ci = 0; ci < Cin; ++ci)
        wp[((size_t)co * K + k) * Cp + ci] = w[((size_t)co * K + k) * Cin + ci];
      w.swap(wp);
    }
    o.w = up(w); o.b = up(b); o.N = Cout; o.Cin = Cp; o.taps = K; o.dil = dil; o.pad = pad; o.stride = stride;
    return true;
  }
  // ConvTransposed1d weight w[co][k][ci], y[t s + k - p] += x[t] w[.][k][.]: output phase r = (t_out + p) mod s is a GEMM over the
  // window x[t1 - (nt-1) .. t1], t1 = (t_out + p) div s, whose tap q multiplies kernel index r + (nt-1-q) s (zero beyond K)
  bool convt(const std::string& p, int Cout, int K, int Cin, int s, int pad, Conv& o) {
    std::vector<float> w, b;
    if (!f32(p + ".weight", w, {Cout, K, Cin}) || !f32(p + ".bias", b, {Cout})) return false;
    const int nt = (K + s - 1) / s;
    std::vector<float> ph((size_t)s * Cout * nt * Cin, 0.f);
    for (int r = 0; r < s; ++r) for (int co = 0; co < Cout; ++co) for (int q = 0; q < nt; ++q) {
      const int k = r + (nt - 1 - q) * s;
      if (k >= K) continue;
      for (int ci = 0; ci < Cin; ++ci) ph[(((size_t)r * Cout + co) * nt + q) * Cin + ci] = w[((size_t)co * K + k) * Cin + ci];
    }
    o.w = up(ph); o.b = up(b); o.N = Cout; o.Cin = Cin; o.taps = nt; o.stride = s; o.pad = pad; o.K = K;
    return true;
  }
  // Snake alpha [C] -> (alpha, 1/clamped alpha)   (HiFiGAN.swift:53-66)
  bool snake(const std::string& n, int C, Conv& o) {
    std::vector<float> a;
    if (!f32(n, a, {C})) return false;
    std::vector<float> ra(C);
    for (int i = 0; i < C; ++i) {
      const float ab = std::fabs(a[i]);
      float cl = (a[i] > 0.f ? 1.f : (a[i] < 0.f ? -1.f : 0.f)) * std::max(ab, 1e-4f);
      if (ab < 1e-9f) cl = 1e-4f;
      ra[i] = 1.0f / cl;
    }
    o.alpha = up(a); o.ralpha = up(ra);
    return true;
  }
  bool resblock(const std::string& p, int C, int k, ResBlock& rb) {
    const mia_hift_config& g = c->cfg;
    for (int i = 0; i < g.n_dilations; ++i) {
      const int d = g.dilations[i];
      const std::string si = std::to_string(i);
      if (!conv(p + ".convs1." + si, C, k, C, d, (k * d - d) / 2, 1, rb.c1[i])) return false;
      if (!conv(p + ".convs2." + si, C, k, C, 1, (k - 1) / 2, 1, rb.c2[i])) return false;
      if (!snake(p + ".activations1." + si + ".alpha", C, rb.c1[i])) return false;
      if (!snake(p + ".activations2." + si + ".alpha", C, rb.c2[i])) return false;
    }
    return true;
  }
};

struct ConvOpt {
  const float* R = nullptr; const float* R2 = nullptr; float scale = 0.f; float lrelu = 0.f; int act = 0;
  int ldy = 0;   // 0 = N
};

// Stacked utterances (mia_hift_vocode_batch): U sequences side by side in every buffer, sequence u at row u * (rows per sequence) of
// its buffer; len[u] = its valid rows in the convolution's INPUT (the rest of its rows read as zero, like the padding past a single
// utterance's end).  U = 1 (every single-utterance entry point): the launches are exactly the unstacked ones.
struct Seq { int U = 1; const int32_t* len = nullptr; };

void seq_args(ConvGemmArgs& g, const Seq& q, int64_t x_rows, int64_t y_rows) {
  if (q.U <= 1) return;
  g.n_seq = q.U; g.x_seq_step = x_rows; g.y_seq_step = y_rows; g.seq_len = q.len;
}

int run_conv(mia_hift* h, const Conv& c, const float* X, int64_t T_in, float* Y, int64_t T_out, const ConvOpt& o = ConvOpt(), const Seq& q = Seq()) {
  ConvGemmArgs g;
  g.X = X; g.ldx = c.Cin; g.T_in = (int)T_in; g.W = c.w; g.bias = c.b; g.alpha = c.alpha; g.ralpha = c.ralpha; g.lrelu_slope = o.lrelu;
  g.Y = Y; g.ldy = o.ldy ? o.ldy : c.N; g.T_out = (int)T_out; g.R = o.R; g.R2 = o.R2; g.ldr = c.N; g.out_scale = o.scale;
  g.M = (int)T_out; g.N = c.N; g.Cin = c.Cin; g.taps = c.taps; g.dil = c.dil; g.pad = c.pad; g.x_row_mul = c.stride; g.gelu = o.act;
  seq_args(g, q, T_in, T_out);
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(h->ctx, MIA_ERR_INVALID_ARGUMENT, "hift: %s", e);
  if (codec_conv_gemm_launch(g, 1, h->ctx->stream)) return mia_fail(h->ctx, MIA_ERR_DEVICE, "hift: conv launch failed");
  return MIA_OK;
}

// transposed conv: Y rows [row_shift, row_shift + T_out) of a buffer with T_out + row_shift rows (per sequence)
int run_convt(mia_hift* h, const Conv& c, const float* X, int64_t T_in, float* Y, int64_t T_out, int row_shift, float lrelu, const Seq& q = Seq()) {
  ConvGemmArgs g;
  g.X = X; g.ldx = c.Cin; g.T_in = (int)T_in; g.W = c.w; g.w_phase_stride = (int64_t)c.N * c.taps * c.Cin; g.bias = c.b; g.lrelu_slope = lrelu;
  g.M = (int)T_in + c.taps - 1; g.N = c.N; g.Cin = c.Cin; g.taps = c.taps; g.dil = 1; g.pad = c.taps - 1;
  g.Y = Y + (int64_t)row_shift * c.N; g.ldy = c.N; g.T_out = (int)T_out; g.y_row_mul = c.stride; g.y_row_off = -c.pad; g.y_phase_step = 1;
  seq_args(g, q, T_in, T_out + row_shift);
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(h->ctx, MIA_ERR_INVALID_ARGUMENT, "hift: %s", e);
  if (codec_conv_gemm_launch(g, c.stride, h->ctx->stream)) return mia_fail(h->ctx, MIA_ERR_DEVICE, "hift: convT launch failed");
  return MIA_OK;
}

// HiFiGANResBlock (HiFiGAN.swift:117-130).  x: input (left untouched), xt: scratch, res: running result; the last pair writes
// out = scale * (conv2(..) + res) + R2 (scale 0 = plain) so that means and sums fold into the final epilogue.
int run_resblock(mia_hift* h, const ResBlock& rb, const float* x, float* xt, float* res, float* out, int64_t T, float scale, const float* R2,
                 const Seq& q = Seq()) {
  const int nd = h->cfg.n_dilations;
  for (int i = 0; i < nd; ++i) {
    const float* in = i == 0 ? x : res;
    if (int rc = run_conv(h, rb.c1[i], in, T, xt, T, ConvOpt(), q)) return rc;
    ConvOpt o; o.R = in;
    float* dst = res;
    if (i == nd - 1) { dst = out; o.scale = scale; o.R2 = R2; }
    if (int rc = run_conv(h, rb.c2[i], xt, T, dst, T, o, q)) return rc;
  }
  return MIA_OK;
}

// p.T / L / F / rows: the LONGEST utterance of the call (every buffer holds U sequences of that many rows); Tu / Lu / Fu: each one's own
struct Plan {
  int64_t T, L, F;
  int64_t rows[4]; int ch[4];
  size_t big;   // floats per ping-pong buffer and sequence
  int U = 1;
  std::vector<int64_t> Tu, Lu, Fu;
  // device [U] valid-row tables of the stacked call (null at U = 1): mel frames, STFT frames, rows after upsample stage i
  const int32_t* len_T = nullptr; const int32_t* len_F = nullptr; const int32_t* len_rows[4] = {nullptr, nullptr, nullptr, nullptr};
};

int64_t rows_after(const mia_hift* h, int64_t T, int stage) {
  int64_t r = T;
  for (int i = 0; i <= stage; ++i) {
    const Conv& u = h->ups[i];
    r = (r - 1) * u.stride - 2 * u.pad + u.K;
    if (i == h->cfg.n_ups - 1) r += 1;
  }
  return r;
}

Plan make_plan(const mia_hift* h, int T) {
  Plan p; p.T = T; p.L = (int64_t)T * h->up; p.F = p.L / 4 + 1;
  size_t big = (size_t)T * h->cfg.base_channels;
  for (int i = 0; i < h->cfg.n_ups; ++i) {
    const Conv& u = h->ups[i];
    p.rows[i] = rows_after(h, T, i); p.ch[i] = u.N;
    big = std::max(big, (size_t)p.rows[i] * u.N);
  }
  p.big = big + 64;
  p.Tu = {p.T}; p.Lu = {p.L}; p.Fu = {p.F};
  return p;
}

struct Scratch {
  float *mel_raw, *melp, *fa, *fb, *f0, *P, *s, *noise, *cache, *stft, *post, *fr, *pcm, *big[4];
  int32_t* lens = nullptr;      // stacked call: the plan's valid-row tables
};

int carve(mia_hift* h, const Plan& p, Scratch& sc) {
  const int C = h->cfg.in_channels, B = h->cfg.base_channels;
  const size_t U = (size_t)p.U;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  size_t sizes[] = {al(U * C * p.T), al(U * h->Cp_mel * p.T), al(U * B * p.T), al(U * B * p.T), al(U * p.T), al(U * p.T * h->H),
                    al(U * p.L), al(U * p.L * h->H), al(p.L), al(U * p.F * 32), al(U * p.F * 18), al(U * p.F * 16), al(U * p.L),
                    al(U * p.big), al(U * p.big), al(U * p.big), al(U * p.big), al(U * 8)};
  size_t tot = 0;
  for (size_t s : sizes) tot += s;
  if (tot > h->arena_floats) {
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->arena) (void)hipFree(h->arena);
    h->arena = nullptr; h->arena_floats = 0;
    if (hipMalloc((void**)&h->arena, tot * 4) != hipSuccess) return mia_fail(h->ctx, MIA_ERR_OUT_OF_MEMORY, "hift: scratch hipMalloc failed");
    h->arena_floats = tot;
  }
  float* q = h->arena;
  float** dst[] = {&sc.mel_raw, &sc.melp, &sc.fa, &sc.fb, &sc.f0, &sc.P, &sc.s, &sc.noise, &sc.cache, &sc.stft, &sc.post, &sc.fr, &sc.pcm,
                   &sc.big[0], &sc.big[1], &sc.big[2], &sc.big[3]};
  for (int i = 0; i < 17; ++i) { *dst[i] = q; q += sizes[i]; }
  sc.lens = reinterpret_cast<int32_t*>(q);
  return MIA_OK;
}

int download(mia_hift* h, float* dst, const float* src, size_t n) {
  MIA_HIP(h->ctx, hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToHost, h->ctx->stream));
  MIA_HIP(h->ctx, hipStreamSynchronize(h->ctx->stream));
  return MIA_OK;
}

// mel of sequence u: [C][Tu] at mel + C * (sum of the T before it); packed to [u][p.T][Cp] (single call: u = 0, offset 0)
int upload_mel(mia_hift* h, const Plan& p, Scratch& sc, const float* mel, bool dev) {
  hipStream_t s = h->ctx->stream;
  const int C = h->cfg.in_channels;
  int64_t tot = 0;
  for (int64_t t : p.Tu) tot += t;
  if (dev) sc.mel_raw = const_cast<float*>(mel);
  else MIA_HIP(h->ctx, hipMemcpyAsync(sc.mel_raw, mel, (size_t)C * tot * 4, hipMemcpyHostToDevice, s));
  if (p.U > 1) MIA_HIP(h->ctx, hipMemsetAsync(sc.melp, 0, (size_t)p.U * p.T * h->Cp_mel * 4, s));      // rows past an utterance's end
  int64_t off = 0;
  for (int u = 0; u < p.U; ++u) {
    dim3 grid((unsigned)((p.Tu[u] + 31) / 32), (unsigned)(h->Cp_mel / 32));
    hipLaunchKernelGGL(hift_pack_mel, grid, dim3(256), 0, s, sc.mel_raw + (int64_t)C * off, sc.melp + (int64_t)u * p.T * h->Cp_mel, C, h->Cp_mel, (int)p.Tu[u]);
    off += p.Tu[u];
  }
  return MIA_OK;
}

int dev_f0(mia_hift* h, const Plan& p, const Scratch& sc) {
  ConvOpt elu; elu.act = 2;
  Seq q; q.U = p.U; q.len = p.len_T;
  const float* x = sc.melp; float* a = sc.fa; float* b = sc.fb;
  for (int i = 0; i < 5; ++i) {
    if (int rc = run_conv(h, h->f0c[i], x, p.T, a, p.T, elu, q)) return rc;
    x = a; std::swap(a, b);
  }
  ConvOpt ab; ab.act = 3; ab.ldy = 1;
  return run_conv(h, h->cls, x, p.T, sc.f0, p.T, ab, q);
}

// noise of sequence u: [Lu][H] at noise + H * (sum of the L before it); source [u][p.L]
int dev_source(mia_hift* h, const Plan& p, const Scratch& sc, bool have_noise, int cache_len) {
  hipStream_t s = h->ctx->stream;
  const mia_hift_config& g = h->cfg;
  int64_t noff = 0;
  for (int u = 0; u < p.U; ++u) {
    const int64_t T = p.Tu[u], L = p.Lu[u];
    const float* f0 = sc.f0 + (int64_t)u * p.T;
    float* P = sc.P + (int64_t)u * p.T * h->H;
    hipLaunchKernelGGL(hift_phase_scan, dim3(1), dim3(256), 0, s, f0, P, (int)T, h->H, (float)g.sampling_rate, (float)h->up);
    const float ratio = (float)T / (float)L;
    const float clip_hi = (float)T - 1.001f;
    hipLaunchKernelGGL(hift_source_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, f0, P, have_noise ? sc.noise + noff * h->H : nullptr,
                       h->lw, h->lb, sc.cache, cache_len, sc.s + (int64_t)u * p.L, (int)T, h->up, h->H, ratio, clip_hi, g.nsf_alpha, g.nsf_sigma,
                       g.voiced_threshold);
    noff += L;
  }
  return hipGetLastError() == hipSuccess ? MIA_OK : mia_fail(h->ctx, MIA_ERR_DEVICE, "hift: source launch failed");
}

int dev_decode(mia_hift* h, const Plan& p, const Scratch& sc) {
  hipStream_t s = h->ctx->stream;
  const mia_hift_config& g = h->cfg;
  const int U = p.U;
  if (U > 1) MIA_HIP(h->ctx, hipMemsetAsync(sc.stft, 0, (size_t)U * p.F * 32 * 4, s));                  // frames past an utterance's end
  for (int u = 0; u < U; ++u)
    hipLaunchKernelGGL(hift_stft, dim3((unsigned)((p.Fu[u] + 255) / 256)), dim3(256), 0, s, sc.s + (int64_t)u * p.L, sc.stft + (int64_t)u * p.F * 32, p.Lu[u], p.Fu[u]);
  float* A = sc.big[0]; float* B = sc.big[1]; float* C = sc.big[2]; float* D = sc.big[3];
  Seq qT; qT.U = U; qT.len = p.len_T;
  Seq qF; qF.U = U; qF.len = p.len_F;
  if (int rc = run_conv(h, h->pre, sc.melp, p.T, A, p.T, ConvOpt(), qT)) return rc;
  int64_t rows = p.T;
  Seq qin = qT;
  for (int i = 0; i < g.n_ups; ++i) {
    const bool last = i == g.n_ups - 1;
    const int64_t r_out = p.rows[i]; const int Ci = p.ch[i];
    Seq qr; qr.U = U; qr.len = p.len_rows[i];
    // leaky-ReLU -> ConvTransposed1d (-> prepend the reflected sample: new[0] = old[1] = new[2])
    if (int rc = run_convt(h, h->ups[i], A, rows, D, last ? r_out - 1 : r_out, last ? 1 : 0, g.lrelu_slope, qin)) return rc;
    if (last) MIA_HIP(h->ctx, hipMemcpy2DAsync(D, (size_t)r_out * Ci * 4, D + 2 * (int64_t)Ci, (size_t)r_out * Ci * 4, (size_t)Ci * 4, (size_t)U, hipMemcpyDeviceToDevice, s));
    std::swap(A, D);
    rows = r_out;
    // source fusion: h += source_resblock(source_down(stft))
    const Conv& sd = h->sdown[i];
    const int64_t sd_rows = (p.F + 2 * sd.pad - sd.taps) / sd.stride + 1;
    if (sd_rows != rows) return mia_fail(h->ctx, MIA_ERR_INVALID_ARGUMENT, "hift: source_downs.%d yields %lld rows, the upsampled stream has %lld", i, (long long)sd_rows, (long long)rows);
    if (int rc = run_conv(h, sd, sc.stft, p.F, C, rows, ConvOpt(), qF)) return rc;
    if (int rc = run_resblock(h, h->srb[i], C, B, C, A, rows, 1.0f, A, qr)) return rc;
    // mean of the residual blocks
    const float inv = 1.0f / (float)g.n_res_kernels;
    for (int k = 0; k < g.n_res_kernels; ++k)
      if (int rc = run_resblock(h, h->rb[i][k], A, B, C, D, rows, inv, k == 0 ? nullptr : D, qr)) return rc;
    std::swap(A, D);
    qin = qr;
  }
  ConvOpt o; o.lrelu = 0.01f;   // default negative_slope before conv_post (CosyHiFTGenerator.swift:445-446)
  if (int rc = run_conv(h, h->post, A, rows, sc.post, rows, o, qin)) return rc;
  for (int u = 0; u < U; ++u) {
    const int64_t F = p.Fu[u], L = p.Lu[u];
    hipLaunchKernelGGL(hift_istft_frames, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, sc.post + (int64_t)u * rows * 18, sc.fr + (int64_t)u * p.F * 16, F);
    hipLaunchKernelGGL(hift_overlap_add, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, sc.fr + (int64_t)u * p.F * 16, sc.pcm + (int64_t)u * p.L, F, L, g.audio_limit);
  }
  return hipGetLastError() == hipSuccess ? MIA_OK : mia_fail(h->ctx, MIA_ERR_DEVICE, "hift: istft launch failed");
}

int check_T(mia_hift* h, int T) {
  if (!h) return MIA_ERR_INVALID_ARGUMENT;
  if (T < 2 || T > 17000) return mia_fail(h->ctx, MIA_ERR_INVALID_ARGUMENT, "hift: T must be in [2, 17000] mel frames (got %d)", T);
  return MIA_OK;
}

}  // namespace

extern "C" {

mia_hift* mia_hift_load(mia_ctx* ctx, const mia_hift_config* cfg, const mia_tensor_view* tensors, int n_tensors) {
  if (!ctx) return nullptr;
  if (!cfg || !tensors) { mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "hift_load: null argument"); return nullptr; }
  const mia_hift_config& g = *cfg;
  if (g.n_ups < 1 || g.n_ups > 4 || g.n_res_kernels < 1 || g.n_res_kernels > 4 || g.n_dilations < 1 || g.n_dilations > 4 ||
      g.in_channels < 1 || g.base_channels % (32 << g.n_ups) || g.nb_harmonics < 0 || g.nb_harmonics > 31) {
    mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "hift_load: unsupported configuration");
    return nullptr;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) { mia_fail(ctx, MIA_ERR_DEVICE, "hift_load: hipSetDevice failed"); return nullptr; }
  mia_hift* h = new mia_hift();
  h->ctx = ctx; h->cfg = g; h->H = g.nb_harmonics + 1; h->Cp_mel = (g.in_channels + 31) / 32 * 32;
  h->up = 4; for (int i = 0; i < g.n_ups; ++i) h->up *= g.up_rates[i];
  HLoader L; L.c = h;
  for (int i = 0; i < n_tensors; ++i) if (tensors[i].name) L.by_name[tensors[i].name] = &tensors[i];
  const int B = g.base_channels;
  bool ok = true;
  static const int cond_idx[5] = {0, 2, 4, 6, 8};
  for (int i = 0; i < 5 && ok; ++i)
    ok = L.conv("f0_predictor.condnet_" + std::to_string(cond_idx[i]), B, 3, i == 0 ? g.in_channels : B, 1, 1, 1, h->f0c[i]);
  if (ok) {   // classifier Linear(B, 1): weight [1][B] == a k=1 convolution
    std::vector<float> w, b;
    ok = L.f32("f0_predictor.classifier.weight", w, {1, B}) && L.f32("f0_predictor.classifier.bias", b, {1});
    if (ok) { h->cls.w = L.up(w); h->cls.b = L.up(b); h->cls.N = 1; h->cls.Cin = B; h->cls.taps = 1; }
  }
  if (ok) {
    std::vector<float> w, b;
    ok = L.f32("m_source.l_linear.weight", w, {1, h->H}) && L.f32("m_source.l_linear.bias", b, {1});
    if (ok) { h->lw = L.up(w); h->lb = b[0]; }
  }
  ok = ok && L.conv("conv_pre", B, 7, g.in_channels, 1, 3, 1, h->pre);
  // cumulative down rates of the source path (CosyHiFTGenerator.swift:349-382): stage i sees the STFT at hop prod(up_rates[i+1:])
  for (int i = 0; i < g.n_ups && ok; ++i) {
    const int Cin = B >> i, Cout = B >> (i + 1);
    const int k = g.up_kernels[i], u = g.up_rates[i];
    if (k < u || (k - u) % 2) { L.err = "hift_load: upsample kernel/stride pair not supported"; ok = false; break; }
    ok = L.convt("ups." + std::to_string(i), Cout, k, Cin, u, (k - u) / 2, h->ups[i]);
    int dr = 1; for (int j = i + 1; j < g.n_ups; ++j) dr *= g.up_rates[j];
    if (ok) ok = dr == 1 ? L.conv("source_downs." + std::to_string(i), Cout, 1, 18, 1, 0, 1, h->sdown[i])
                         : L.conv("source_downs." + std::to_string(i), Cout, 2 * dr, 18, 1, dr / 2, dr, h->sdown[i]);
    if (ok) ok = L.resblock("source_resblocks." + std::to_string(i), Cout, g.src_res_kernels[i], h->srb[i]);
    for (int k2 = 0; k2 < g.n_res_kernels && ok; ++k2)
      ok = L.resblock("resblocks." + std::to_string(i * g.n_res_kernels + k2), Cout, g.res_kernels[k2], h->rb[i][k2]);
  }
  ok = ok && L.conv("conv_post", 18, 7, B >> g.n_ups, 1, 3, 1, h->post);
  if (!ok || !L.err.empty()) {
    mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "hift_load: %s", L.err.empty() ? "failed" : L.err.c_str());
    mia_hift_free(h);
    return nullptr;
  }
  return h;
}

void mia_hift_free(mia_hift* h) {
  if (!h) return;
  (void)hipStreamSynchronize(h->ctx->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->arena) (void)hipFree(h->arena);
  delete h;
}

int mia_hift_upsample_factor(const mia_hift* h) { return h ? h->up : 0; }

int mia_hift_f0(mia_hift* h, const float* mel, int T, float* f0, int mem) {
  if (int rc = check_T(h, T)) return rc;
  MIA_CHECK_ARG(h->ctx, mel && f0 && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "hift_f0: bad argument");
  MIA_HIP(h->ctx, hipSetDevice(h->ctx->device));
  Plan p = make_plan(h, T); Scratch sc;
  if (int rc = carve(h, p, sc)) return rc;
  const bool dev = mem == MIA_MEM_DEVICE;
  if (dev) sc.f0 = f0;
  if (int rc = upload_mel(h, p, sc, mel, dev)) return rc;
  if (int rc = dev_f0(h, p, sc)) return rc;
  return dev ? MIA_OK : download(h, f0, sc.f0, (size_t)T);
}

int mia_hift_source(mia_hift* h, const float* f0, int T, const float* noise, float* source, int mem) {
  if (int rc = check_T(h, T)) return rc;
  MIA_CHECK_ARG(h->ctx, f0 && source && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "hift_source: bad argument");
  MIA_HIP(h->ctx, hipSetDevice(h->ctx->device));
  Plan p = make_plan(h, T); Scratch sc;
  if (int rc = carve(h, p, sc)) return rc;
  hipStream_t s = h->ctx->stream;
  const bool dev = mem == MIA_MEM_DEVICE;
  if (dev) { sc.f0 = const_cast<float*>(f0); sc.noise = const_cast<float*>(noise); sc.s = source; }
  else {
    MIA_HIP(h->ctx, hipMemcpyAsync(sc.f0, f0, (size_t)T * 4, hipMemcpyHostToDevice, s));
    if (noise) MIA_HIP(h->ctx, hipMemcpyAsync(sc.noise, noise, (size_t)p.L * h->H * 4, hipMemcpyHostToDevice, s));
  }
  if (int rc = dev_source(h, p, sc, noise != nullptr, 0)) return rc;
  return dev ? MIA_OK : download(h, source, sc.s, (size_t)p.L);
}

int mia_hift_decode(mia_hift* h, const float* mel, int T, const float* source, float* pcm, int mem) {
  if (int rc = check_T(h, T)) return rc;
  MIA_CHECK_ARG(h->ctx, mel && source && pcm && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "hift_decode: bad argument");
  MIA_HIP(h->ctx, hipSetDevice(h->ctx->device));
  Plan p = make_plan(h, T); Scratch sc;
  if (int rc = carve(h, p, sc)) return rc;
  hipStream_t s = h->ctx->stream;
  const bool dev = mem == MIA_MEM_DEVICE;
  if (int rc = upload_mel(h, p, sc, mel, dev)) return rc;
  if (dev) { sc.s = const_cast<float*>(source); sc.pcm = pcm; }
  else MIA_HIP(h->ctx, hipMemcpyAsync(sc.s, source, (size_t)p.L * 4, hipMemcpyHostToDevice, s));
  if (int rc = dev_decode(h, p, sc)) return rc;
  return dev ? MIA_OK : download(h, pcm, sc.pcm, (size_t)p.L);
}

int mia_hift_vocode(mia_hift* h, const float* mel, int T, const float* noise, const float* cache_source, int cache_len, float* pcm,
                    float* source_out, int mem) {
  if (int rc = check_T(h, T)) return rc;
  MIA_CHECK_ARG(h->ctx, mel && pcm && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "hift_vocode: bad argument");
  MIA_HIP(h->ctx, hipSetDevice(h->ctx->device));
  Plan p = make_plan(h, T); Scratch sc;
  MIA_CHECK_ARG(h->ctx, cache_len >= 0 && cache_len <= p.L && (cache_len == 0 || cache_source), "hift_vocode: bad cache_source");
  if (int rc = carve(h, p, sc)) return rc;
  hipStream_t s = h->ctx->stream;
  const bool dev = mem == MIA_MEM_DEVICE;
  if (int rc = upload_mel(h, p, sc, mel, dev)) return rc;
  if (dev) {
    sc.noise = const_cast<float*>(noise); sc.cache = const_cast<float*>(cache_source); sc.pcm = pcm;
    if (source_out) sc.s = source_out;
  } else {
    if (noise) MIA_HIP(h->ctx, hipMemcpyAsync(sc.noise, noise, (size_t)p.L * h->H * 4, hipMemcpyHostToDevice, s));
    if (cache_len) MIA_HIP(h->ctx, hipMemcpyAsync(sc.cache, cache_source, (size_t)cache_len * 4, hipMemcpyHostToDevice, s));
  }
  if (int rc = dev_f0(h, p, sc)) return rc;
  if (int rc = dev_source(h, p, sc, noise != nullptr, cache_len)) return rc;
  if (int rc = dev_decode(h, p, sc)) return rc;
  if (dev) return MIA_OK;
  if (source_out) MIA_HIP(h->ctx, hipMemcpyAsync(source_out, sc.s, (size_t)p.L * 4, hipMemcpyDeviceToHost, s));
  return download(h, pcm, sc.pcm, (size_t)p.L);
}

// U utterances in one pass: every convolution of the f0 predictor and of the decoder runs once over the stacked sequences (grid.z =
// sequence x phase of the tap GEMM; a sequence's rows past its own end read as zero, exactly the padding a single call sees), the
// per-sample stages (phase scan, sine source, STFT, iSTFT, overlap-add) run per utterance.  mels: [C][T_u] blocks back to back; noise
// (optional): [L_u][H] blocks back to back, L_u = T_u * upsample_factor; pcm: L_u samples per utterance back to back.  Every utterance
// equals its own mia_hift_vocode call bit for bit (tests/test_hift_gpu.py).  The cache_source hand-over of the streaming path stays
// with the single-utterance call.
int mia_hift_vocode_batch(mia_hift* h, const float* mels, const int32_t* T, int n, const float* noise, float* pcm, int mem) {
  if (!h) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(h->ctx, mels && T && pcm && n >= 1 && n <= 64 && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "hift_vocode_batch: bad argument (1 <= n <= 64)");
  int Tm = 0;
  for (int u = 0; u < n; ++u) { if (int rc = check_T(h, T[u])) return rc; Tm = std::max(Tm, T[u]); }
  MIA_HIP(h->ctx, hipSetDevice(h->ctx->device));
  hipStream_t s = h->ctx->stream;
  Plan p = make_plan(h, Tm); Scratch sc;
  p.U = n; p.Tu.clear(); p.Lu.clear(); p.Fu.clear();
  int64_t Ltot = 0;
  for (int u = 0; u < n; ++u) { const int64_t L = (int64_t)T[u] * h->up; p.Tu.push_back(T[u]); p.Lu.push_back(L); p.Fu.push_back(L / 4 + 1); Ltot += L; }
  if (int rc = carve(h, p, sc)) return rc;
  const bool dev = mem == MIA_MEM_DEVICE;
  std::vector<int32_t> lens((size_t)(2 + h->cfg.n_ups) * n);
  for (int u = 0; u < n; ++u) {
    lens[u] = T[u]; lens[(size_t)n + u] = (int32_t)p.Fu[u];
    for (int i = 0; i < h->cfg.n_ups; ++i) lens[(size_t)(2 + i) * n + u] = (int32_t)rows_after(h, T[u], i);
  }
  if (n > 1) {
    MIA_HIP(h->ctx, hipMemcpyAsync(sc.lens, lens.data(), lens.size() * 4, hipMemcpyHostToDevice, s));
    p.len_T = sc.lens; p.len_F = sc.lens + n;
    for (int i = 0; i < h->cfg.n_ups; ++i) p.len_rows[i] = sc.lens + (size_t)(2 + i) * n;
  }
  if (int rc = upload_mel(h, p, sc, mels, dev)) return rc;
  if (noise) {
    if (dev) sc.noise = const_cast<float*>(noise);
    else MIA_HIP(h->ctx, hipMemcpyAsync(sc.noise, noise, (size_t)Ltot * h->H * 4, hipMemcpyHostToDevice, s));
  }
  if (int rc = dev_f0(h, p, sc)) return rc;
  if (int rc = dev_source(h, p, sc, noise != nullptr, 0)) return rc;
  if (int rc = dev_decode(h, p, sc)) return rc;
  int64_t off = 0;
  for (int u = 0; u < n; ++u) {
    MIA_HIP(h->ctx, hipMemcpyAsync(pcm + off, sc.pcm + (int64_t)u * p.L, (size_t)p.Lu[u] * 4, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    off += p.Lu[u];
  }
  MIA_HIP(h->ctx, hipStreamSynchronize(s));       // (also keeps `lens` alive until its upload has been read)
  return MIA_OK;
}

}  // extern "C"
