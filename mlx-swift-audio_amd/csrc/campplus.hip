// campplus.hip -- CAM++ speaker encoder and its Kaldi filterbank front end: the once-per-speaker part of
// CosyVoice2's prepareConditionals (TTS/CosyVoice2/CosyVoice2TTS.swift:409, SpeakerEncoder/CAMPlusSpeakerEncoder.swift:12-150).
//
// Replaces Codec/S3Gen/CAMPPlus.swift: kaldiFbankCAMPPlus :32-108 (snip-edges frames of 400 @ hop 160, per-frame DC removal,
// 0.97 pre-emphasis, Povey window, 512-point power spectrum, 80 HTK triangles on rounded bin edges, log clamped at FLT_EPSILON),
// CAMPPlus.inference :788-818 (per-bin time-mean removal) and CAMPPlus.callAsFunction :755-785: FCM head (2-D 3x3 convolutions
// over (bin, frame), 32 channels, three stride-2 stages along the bin axis) -> TDNN k5 s2 -> three densely connected blocks
// (12 / 24 / 16 layers of BN-ReLU -> 1x1 -> BN-ReLU -> context-aware-masked dilated conv) with halving transit layers ->
// BN-ReLU -> statistics pooling -> 192-d embedding.  BatchNorm is inference-mode (running statistics, eps 1e-5).
//
// Everything is fp32.  The DFT and every 1-D convolution / 1x1 layer run on the exact-fp32 matrix cores through conv_gemm_f32
// (codec_kernels.hip); the 32-channel 2-D convolutions, the context gate and the pooling are small dedicated kernels -- the whole
// encoder is ~7 M parameters at T <= 1500 frames and runs once per speaker, so the point here is exactness and one code path,
// not a roofline.
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "codec.h"
#include "mia_device.h"
#include "mia_internal.h"
#include "tensor_loader.h"

namespace {

constexpr int FB_WIN = 400, FB_HOP = 160, FB_NFFT = 512, FB_NBIN = 257, FB_NBP = 288, FB_K = 416, FB_NMEL = 80;
constexpr int FCM_C = 32;
constexpr float BN_EPS = 1e-5f;
constexpr int SEG_LEN = 100;

struct Conv2 { float* wk = nullptr; float* scale = nullptr; float* shift = nullptr; int cin = 0, k = 0; };   // wk [k*k*cin][32]
struct Lin { float* w = nullptr; float* bias = nullptr; int cout = 0, cin = 0, taps = 1; };                 // w [cout][taps*cin]
struct Affine { float* scale = nullptr; float* shift = nullptr; };
struct DenseLayer { Affine bn1; Lin lin1; Lin local; float* w1; float* b1; float* w2; float* b2; int cin, dil; };

}  // namespace

struct mia_campplus {
  mia_ctx* ctx = nullptr;
  std::vector<void*> allocs;
  float* dft = nullptr; float* window = nullptr; float* fb_w = nullptr; int* fb_meta = nullptr;
  Conv2 conv1, conv2;
  struct Res { Conv2 c1, c2, sc; int stride; } res[4];
  Lin tdnn;
  std::vector<DenseLayer> layers[3];
  Affine tbn[3]; Lin tlin[3];
  Affine out_bn;
  float* dense_w = nullptr; float* dense_scale = nullptr; float* dense_shift = nullptr;   // [192][1024], BN without affine folded to scale/shift
};

namespace {

// ---- Kaldi filterbank ---------------------------------------------------------------------------------------------------------------
// one workgroup per frame: DC removal, pre-emphasis inside the frame, Povey window, zero pad to 416 (13 rows of 32 for the DFT GEMM)
__global__ __launch_bounds__(256) void fbank_frames(const float* __restrict__ x, const float* __restrict__ window, float* __restrict__ frames, int64_t n) {
  __shared__ float fr[FB_WIN];
  __shared__ float sh[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const float* src = x + (int64_t)f * FB_HOP;
  float part = 0.f;
  for (int i = tid; i < FB_WIN; i += 256) { const float v = src[i]; fr[i] = v; part += v; }
  part = wave_sum(part);
  if ((tid & 63) == 0) sh[tid >> 6] = part;
  __syncthreads();
  const float mean = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)FB_WIN;
  float* dst = frames + (int64_t)f * FB_K;
  for (int i = tid; i < FB_K; i += 256) {
    float v = 0.f;
    if (i < FB_WIN) {
      const float cur = fr[i] - mean;
      v = i == 0 ? cur : cur - 0.97f * (fr[i - 1] - mean);
      v *= window[i];
    }
    dst[i] = v;
  }
}

// spec [F][2 NBP] (cos | -sin) -> fb[f][m] = log(max(sum_k |X_k|^2 w[m][k], FLT_EPSILON))
__global__ __launch_bounds__(256) void fbank_finish(const float* __restrict__ spec, const float* __restrict__ fb_w, const int* __restrict__ fb_meta,
                                                    float* __restrict__ out, int F) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)F * FB_NMEL) return;
  const int f = (int)(e / FB_NMEL), m = (int)(e % FB_NMEL);
  const int lo = fb_meta[m * 3], cnt = fb_meta[m * 3 + 1], off = fb_meta[m * 3 + 2];
  const float* re = spec + (int64_t)f * 2 * FB_NBP + lo;
  const float* im = re + FB_NBP;
  float acc = 0.f;
  for (int c = 0; c < cnt; ++c) acc = fmaf(re[c] * re[c] + im[c] * im[c], fb_w[off + c], acc);
  out[e] = logf(fmaxf(acc, 1.1920929e-07f));
}

// fb[f][m] -= mean_f fb[f][m]        (CAMPPlus.inference :797-799); one workgroup per mel bin
__global__ __launch_bounds__(256) void fbank_mean_sub(float* __restrict__ fb, int F) {
  __shared__ float sh[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  float part = 0.f;
  for (int f = tid; f < F; f += 256) part += fb[(int64_t)f * FB_NMEL + m];
  part = wave_sum(part);
  if ((tid & 63) == 0) sh[tid >> 6] = part;
  __syncthreads();
  const float mean = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)F;
  for (int f = tid; f < F; f += 256) fb[(int64_t)f * FB_NMEL + m] -= mean;
}

// ---- FCM: 32-channel 2-D convolution, channels-last [H][W][CIN] like MLX ---------------------------------------------------------------
// y[h][t][co] = act( (sum_{kh,kw,ci} x[h*s + kh - p][t + kw - p][ci] * w[co][kh][kw][ci]) * scale[co] + shift[co] (+ res[h][t][co]) )
// 8 frames x 32 output channels per workgroup; the whole filter sits in LDS k-major so the 32 channel lanes read consecutive words
// and the input value is a broadcast.  Output strides are free: the last FCM layer writes the [T][c*H + h] layout the TDNN reads.
template <int CIN>
__global__ __launch_bounds__(256) void cam_conv2d(const float* __restrict__ x, int H_in, int W, const float* __restrict__ wk, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const float* __restrict__ res, float* __restrict__ y, int K,
                                                  int stride_h, int relu, int64_t ys_h, int64_t ys_t, int64_t ys_c) {
  extern __shared__ float wl[];             // [K*K*CIN][32]
  const int tid = threadIdx.x, co = tid & 31, tl = tid >> 5;
  const int nw = K * K * CIN * FCM_C;
  for (int i = tid; i < nw; i += 256) wl[i] = wk[i];
  __syncthreads();
  const int t = blockIdx.x * 8 + tl, h = blockIdx.y, pad = K / 2;
  if (t >= W) return;
  float acc = 0.f;
  for (int kh = 0; kh < K; ++kh) {
    const int hi = h * stride_h + kh - pad;
    if (hi < 0 || hi >= H_in) continue;
    for (int kw = 0; kw < K; ++kw) {
      const int ti = t + kw - pad;
      if (ti < 0 || ti >= W) continue;
      const float* xp = x + ((int64_t)hi * W + ti) * CIN;
      const float* wp = wl + (kh * K + kw) * CIN * FCM_C + co;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc = fmaf(xp[ci], wp[ci * FCM_C], acc);
    }
  }
  float v = acc * scale[co] + shift[co];
  if (res) v += res[((int64_t)h * W + t) * FCM_C + co];
  if (relu) v = fmaxf(v, 0.f);
  y[h * ys_h + t * ys_t + co * ys_c] = v;
}

// ---- 1-D part: rows = frames, channels contiguous ------------------------------------------------------------------------------------
// y[t][c] = relu(x[t][c] * scale[c] + shift[c])      (the BatchNorm-ReLU in front of every 1x1)
__global__ __launch_bounds__(256) void cam_bn_relu(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int T, int C,
                                                   const float* __restrict__ scale, const float* __restrict__ shift) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)T * C) return;
  const int t = (int)(e / C), c = (int)(e % C);
  y[(int64_t)t * ldy + c] = fmaxf(x[(int64_t)t * ldx + c] * scale[c] + shift[c], 0.f);
}

// CAMLayer's mask (CAMPPlus.swift:470-503): context = mean over all frames + mean over the frame's 100-frame segment (a short last
// segment still divides by 100: the reference zero-pads before averaging); gate = sigmoid(W2 relu(W1 context + b1) + b2).
// The context is constant inside a segment, so the two 1x1 layers run once per segment: one workgroup per segment (8 row groups
// share the frame loop).
__global__ __launch_bounds__(1024) void cam_gate(const float* __restrict__ h, int T, const float* __restrict__ w1, const float* __restrict__ b1,
                                                 const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ gate) {
  __shared__ float ptot[8][128];
  __shared__ float pseg[8][128];
  __shared__ float ctx[128];
  __shared__ float hid[64];
  const int seg = blockIdx.x, c = threadIdx.x & 127, grp = threadIdx.x >> 7;        // 8 row groups x 128 channels
  float tot = 0.f, sg = 0.f;
  const int lo = seg * SEG_LEN, hi = min(T, lo + SEG_LEN);
  for (int t = grp; t < T; t += 8) { const float v = h[(int64_t)t * 128 + c]; tot += v; if (t >= lo && t < hi) sg += v; }
  ptot[grp][c] = tot; pseg[grp][c] = sg;
  __syncthreads();
  if (threadIdx.x < 128) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) { a += ptot[g][c]; b += pseg[g][c]; }
    ctx[c] = a / (float)T + b / (float)SEG_LEN;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float a = b1[c];
    for (int k = 0; k < 128; ++k) a = fmaf(w1[c * 128 + k], ctx[k], a);
    hid[c] = fmaxf(a, 0.f);
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float a = b2[c];
    for (int k = 0; k < 64; ++k) a = fmaf(w2[c * 64 + k], hid[k], a);
    gate[seg * 32 + c] = 1.0f / (1.0f + expf(-a));
  }
}

// dst[t][n] = y[t][n] * gate[t / 100][n]      (written straight into the block's concatenation buffer, 32 new channels)
__global__ __launch_bounds__(256) void cam_apply(const float* __restrict__ y, const float* __restrict__ gate, float* __restrict__ dst, int64_t ldd, int T) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)T * 32) return;
  const int t = (int)(e >> 5), n = (int)(e & 31);
  dst[(int64_t)t * ldd + n] = y[e] * gate[(t / SEG_LEN) * 32 + n];
}

// BN-ReLU then statistics pooling (CAMPPlus.swift:328-333): st[c] = mean_t, st[C + c] = sqrt(var_t + 1e-5) (population variance)
__global__ __launch_bounds__(256) void cam_stats(const float* __restrict__ x, int64_t ldx, int T, int C, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, float* __restrict__ st) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float s = scale[c], b = shift[c];
  float sum = 0.f;
  for (int t = 0; t < T; ++t) sum += fmaxf(x[(int64_t)t * ldx + c] * s + b, 0.f);
  const float mean = sum / (float)T;
  float var = 0.f;
  for (int t = 0; t < T; ++t) { const float d = fmaxf(x[(int64_t)t * ldx + c] * s + b, 0.f) - mean; var = fmaf(d, d, var); }
  st[c] = mean;
  st[C + c] = sqrtf(var / (float)T + 1e-5f);
}

// emb[n] = (sum_k W[n][k] st[k]) * scale[n] + shift[n]     (DenseLayer: 1x1 without bias, BatchNorm without affine)
__global__ __launch_bounds__(64) void cam_dense(const float* __restrict__ st, const float* __restrict__ w, const float* __restrict__ scale,
                                                const float* __restrict__ shift, float* __restrict__ emb, int K) {
  const int n = blockIdx.x, lane = threadIdx.x;
  float a = 0.f;
  for (int k = lane; k < K; k += 64) a = fmaf(w[(int64_t)n * K + k], st[k], a);
  a = wave_sum(a);
  if (lane == 0) emb[n] = a * scale[n] + shift[n];
}

// ---- loading -----------------------------------------------------------------------------------------------------------------------------
struct Loader {
  mia_campplus* m; TensorLoader tl;
  bool bn_fold(const std::string& key, int c, bool affine, std::vector<float>& scale, std::vector<float>& shift) {
    std::vector<float> g, b, rm, rv;
    if (!tl.f32(key + ".running_mean", rm, {c}) || !tl.f32(key + ".running_var", rv, {c})) return false;
    if (affine && (!tl.f32(key + ".weight", g, {c}) || !tl.f32(key + ".bias", b, {c}))) return false;
    scale.resize(c); shift.resize(c);
    for (int i = 0; i < c; ++i) {
      const float inv = 1.0f / sqrtf(rv[i] + BN_EPS);
      scale[i] = affine ? g[i] * inv : inv;
      shift[i] = (affine ? b[i] : 0.f) - rm[i] * scale[i];
    }
    return true;
  }
  bool affine(const std::string& key, int c, Affine& a) {
    std::vector<float> s, h;
    if (!bn_fold(key, c, true, s, h)) return false;
    a.scale = tl.up(s); a.shift = tl.up(h);
    return true;
  }
  // Conv2d [32][k][k][cin] -> k-major [k*k*cin][32]; the BatchNorm that follows stays a separate scale/shift (applied to the fp32 sum)
  bool conv2(const std::string& ckey, const std::string& bkey, int cin, int k, Conv2& c) {
    std::vector<float> w, s, h;
    if (!tl.f32(ckey + ".weight", w, {FCM_C, k, k, cin}) || !bn_fold(bkey, FCM_C, true, s, h)) return false;
    std::vector<float> wk((size_t)k * k * cin * FCM_C);
    for (int o = 0; o < FCM_C; ++o)
      for (int q = 0; q < k * k * cin; ++q) wk[(size_t)q * FCM_C + o] = w[(size_t)o * k * k * cin + q];
    c.wk = tl.up(wk); c.scale = tl.up(s); c.shift = tl.up(h); c.cin = cin; c.k = k;
    return true;
  }
  // Conv1d [cout][taps][cin] is already the GEMM layout; an optional following BatchNorm folds into the rows and becomes the bias
  bool lin(const std::string& key, int cout, int taps, int cin, const std::string& bn_after, bool bias, Lin& l) {
    std::vector<float> w, b;
    if (!tl.f32(key + ".weight", w, {cout, taps, cin})) return false;
    if (bias && !tl.f32(key + ".bias", b, {cout})) return false;
    if (!bn_after.empty()) {
      std::vector<float> s, h;
      if (!bn_fold(bn_after, cout, true, s, h)) return false;
      for (int o = 0; o < cout; ++o) for (int q = 0; q < taps * cin; ++q) w[(size_t)o * taps * cin + q] *= s[o];
      b = h;
    }
    l.w = tl.up(w); l.bias = b.empty() ? nullptr : tl.up(b); l.cout = cout; l.cin = cin; l.taps = taps;
    return true;
  }
};

const int BLOCK_LAYERS[3] = {12, 24, 16};
const int BLOCK_DIL[3] = {1, 2, 2};

int run_lin(mia_ctx* ctx, const Lin& l, const float* X, int64_t ldx, int T_in, float* Y, int64_t ldy, int T_out, int stride, int dil, int pad, int act) {
  ConvGemmArgs g;
  g.X = X; g.ldx = ldx; g.T_in = T_in; g.W = l.w; g.bias = l.bias; g.Y = Y; g.ldy = ldy; g.T_out = T_out;
  g.M = T_out; g.N = l.cout; g.Cin = l.cin; g.taps = l.taps; g.dil = dil; g.pad = pad; g.x_row_mul = stride; g.gelu = act;
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "campplus: %s", e);
  if (codec_conv_gemm_launch(g, 1, ctx->stream)) return mia_fail(ctx, MIA_ERR_DEVICE, "campplus: GEMM launch failed");
  return MIA_OK;
}

void run_conv2(hipStream_t s, const Conv2& c, const float* x, int H_in, int W, const float* res, float* y, int H_out, int stride_h, int relu,
               int64_t ys_h, int64_t ys_t, int64_t ys_c) {
  const dim3 grid((W + 7) / 8, H_out);
  const size_t lds = (size_t)c.k * c.k * c.cin * FCM_C * 4;
  if (c.cin == 1) hipLaunchKernelGGL((cam_conv2d<1>), grid, dim3(256), lds, s, x, H_in, W, c.wk, c.scale, c.shift, res, y, c.k, stride_h, relu, ys_h, ys_t, ys_c);
  else hipLaunchKernelGGL((cam_conv2d<FCM_C>), grid, dim3(256), lds, s, x, H_in, W, c.wk, c.scale, c.shift, res, y, c.k, stride_h, relu, ys_h, ys_t, ys_c);
}

int64_t fbank_frames_of(int64_t n) { return n < FB_WIN ? 0 : (n - FB_WIN) / FB_HOP + 1; }

// audio (device) -> mean-normalised or raw fbank [F][80] (device)
int run_fbank(mia_campplus* m, const float* x, int64_t n, int F, float* frames, float* spec, float* fb, bool mean_norm) {
  mia_ctx* ctx = m->ctx; hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(fbank_frames, dim3(F), dim3(256), 0, s, x, m->window, frames, n);
  ConvGemmArgs g;
  g.X = frames; g.ldx = FB_K; g.T_in = F; g.W = m->dft; g.Y = spec; g.ldy = 2 * FB_NBP; g.T_out = F;
  g.M = F; g.N = 2 * FB_NBP; g.Cin = FB_K; g.taps = 1;
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "campplus fbank: %s", e);
  if (codec_conv_gemm_launch(g, 1, s)) return mia_fail(ctx, MIA_ERR_DEVICE, "campplus fbank: GEMM launch failed");
  hipLaunchKernelGGL(fbank_finish, dim3((unsigned)(((int64_t)F * FB_NMEL + 255) / 256)), dim3(256), 0, s, spec, m->fb_w, m->fb_meta, fb, F);
  if (mean_norm) hipLaunchKernelGGL(fbank_mean_sub, dim3(FB_NMEL), dim3(256), 0, s, fb, F);
  return hipGetLastError() == hipSuccess ? MIA_OK : mia_fail(ctx, MIA_ERR_DEVICE, "campplus fbank: launch failed");
}

}  // namespace

extern "C" void mia_campplus_free(mia_campplus* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
}

extern "C" mia_campplus* mia_campplus_load(mia_ctx* ctx, const mia_tensor_view* tensors, int n_tensors) {
  if (!ctx) return nullptr;
  auto fail = [&](mia_campplus* m, const std::string& msg) -> mia_campplus* { ctx->err = "campplus_load: " + msg; if (m) mia_campplus_free(m); return nullptr; };
  if (!tensors || n_tensors <= 0) return fail(nullptr, "null arguments");
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(nullptr, "hipSetDevice failed");
  mia_campplus* m = new mia_campplus(); m->ctx = ctx;
  Loader L; L.m = m; L.tl.allocs = &m->allocs; L.tl.index(tensors, n_tensors);
  // ---- front-end tables: Povey window, zero-padded 512-point DFT basis restricted to the 400 live samples, HTK triangles
  {
    std::vector<float> win(FB_WIN);
    for (int i = 0; i < FB_WIN; ++i) win[i] = powf(0.5f - 0.5f * cosf(2.0f * (float)M_PI * (float)i / (float)(FB_WIN - 1)), 0.85f);
    m->window = L.tl.up(win);
    std::vector<float> dft((size_t)2 * FB_NBP * FB_K, 0.f);
    for (int k = 0; k < FB_NBIN; ++k)
      for (int n = 0; n < FB_WIN; ++n) {
        const double ang = 2.0 * M_PI * (double)((k * n) % FB_NFFT) / (double)FB_NFFT;
        dft[(size_t)k * FB_K + n] = (float)cos(ang);
        dft[(size_t)(FB_NBP + k) * FB_K + n] = (float)-sin(ang);
      }
    m->dft = L.tl.up(dft);
    auto hz_to_mel = [](float hz) { return 2595.0f * log10f(1.0f + hz / 700.0f); };
    auto mel_to_hz = [](float mel) { return 700.0f * (powf(10.0f, mel / 2595.0f) - 1.0f); };
    const float mel_min = hz_to_mel(20.0f), mel_max = hz_to_mel(8000.0f);
    std::vector<int> bins(FB_NMEL + 2);
    for (int i = 0; i < FB_NMEL + 2; ++i) bins[i] = (int)roundf(mel_to_hz(mel_min + (float)i * (mel_max - mel_min) / (float)(FB_NMEL + 1)) * (float)FB_NFFT / 16000.0f);
    std::vector<float> w; std::vector<int> meta(FB_NMEL * 3);
    for (int q = 1; q <= FB_NMEL; ++q) {
      const int lo = bins[q - 1], c = bins[q], hi = bins[q + 1];
      std::vector<float> row(FB_NBIN, 0.f);
      if (c != lo) for (int k = lo; k < c; ++k) if (k >= 0 && k < FB_NBIN) row[k] = (float)(k - lo) / (float)(c - lo);
      if (hi != c) for (int k = c; k < hi; ++k) if (k >= 0 && k < FB_NBIN) row[k] = (float)(hi - k) / (float)(hi - c);
      int first = -1, last = -2;
      for (int k = 0; k < FB_NBIN; ++k) if (row[k] != 0.f) { if (first < 0) first = k; last = k; }
      if (first < 0) { first = 0; last = -1; }
      meta[(q - 1) * 3] = first; meta[(q - 1) * 3 + 1] = last - first + 1; meta[(q - 1) * 3 + 2] = (int)w.size();
      for (int k = first; k <= last; ++k) w.push_back(row[k]);
    }
    if (w.empty()) w.push_back(0.f);
    m->fb_w = L.tl.up(w);
    std::vector<float> metaf(meta.size());
    memcpy(metaf.data(), meta.data(), meta.size() * 4);
    m->fb_meta = (int*)L.tl.up(metaf);
  }
  // ---- FCM
  L.conv2("head.conv1", "head.bn1", 1, 3, m->conv1);
  const char* rn[4] = {"head.layer1.0", "head.layer1.1", "head.layer2.0", "head.layer2.1"};
  for (int i = 0; i < 4; ++i) {
    const std::string p = rn[i];
    m->res[i].stride = (i % 2 == 0) ? 2 : 1;
    L.conv2(p + ".conv1", p + ".bn1", FCM_C, 3, m->res[i].c1);
    L.conv2(p + ".conv2", p + ".bn2", FCM_C, 3, m->res[i].c2);
    if (m->res[i].stride != 1) L.conv2(p + ".shortcut.0", p + ".shortcut.1", FCM_C, 1, m->res[i].sc);
  }
  L.conv2("head.conv2", "head.bn2", FCM_C, 3, m->conv2);
  // ---- TDNN + dense blocks
  int ch = FCM_C * (FB_NMEL / 8);
  L.lin("tdnn.linear", 128, 5, ch, "tdnn.nonlinear.0", false, m->tdnn);
  ch = 128;
  for (int b = 0; b < 3 && L.tl.err.empty(); ++b) {
    m->layers[b].resize(BLOCK_LAYERS[b]);
    for (int i = 0; i < BLOCK_LAYERS[b] && L.tl.err.empty(); ++i) {
      const std::string p = "blocks." + std::to_string(b) + ".layers." + std::to_string(i);
      DenseLayer& d = m->layers[b][i];
      d.cin = ch + 32 * i; d.dil = BLOCK_DIL[b];
      L.affine(p + ".nonlinear1.0", d.cin, d.bn1);
      L.lin(p + ".linear1", 128, 1, d.cin, p + ".nonlinear2.0", false, d.lin1);
      L.lin(p + ".cam_layer.linear_local", 32, 3, 128, "", false, d.local);
      std::vector<float> t;
      if (L.tl.f32(p + ".cam_layer.linear1.weight", t, {64, 1, 128})) d.w1 = L.tl.up(t);
      if (L.tl.f32(p + ".cam_layer.linear1.bias", t, {64})) d.b1 = L.tl.up(t);
      if (L.tl.f32(p + ".cam_layer.linear2.weight", t, {32, 1, 64})) d.w2 = L.tl.up(t);
      if (L.tl.f32(p + ".cam_layer.linear2.bias", t, {32})) d.b2 = L.tl.up(t);
    }
    ch += 32 * BLOCK_LAYERS[b];
    const std::string tp = "transits." + std::to_string(b);
    L.affine(tp + ".nonlinear.0", ch, m->tbn[b]);
    L.lin(tp + ".linear", ch / 2, 1, ch, "", false, m->tlin[b]);
    ch /= 2;
  }
  L.affine("out_nonlinear.0", ch, m->out_bn);
  {
    std::vector<float> w, s, h;
    if (L.tl.f32("dense.linear.weight", w, {MIA_CAMPPLUS_DIM, 1, 2 * ch}) && L.bn_fold("dense.nonlinear.0", MIA_CAMPPLUS_DIM, false, s, h)) {
      m->dense_w = L.tl.up(w); m->dense_scale = L.tl.up(s); m->dense_shift = L.tl.up(h);
    }
  }
  if (!L.tl.err.empty()) return fail(m, L.tl.err);
  if (hipDeviceSynchronize() != hipSuccess) return fail(m, "device error during upload");
  return m;
}

extern "C" int64_t mia_kaldi_fbank_frames(int64_t n_samples) { return fbank_frames_of(n_samples); }

// kaldiFbankCAMPPlus: pcm 16 kHz [n_samples] -> fbank [frames][80]; mean_norm != 0 subtracts each bin's time mean (the encoder's input)
extern "C" int mia_campplus_fbank(mia_campplus* m, const float* pcm, int64_t n_samples, int mean_norm, float* fbank, int mem) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, pcm && fbank && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "campplus_fbank: bad argument");
  MIA_CHECK_ARG(ctx, n_samples >= FB_WIN && n_samples <= (int64_t)16000 * 600, "campplus_fbank: n_samples must be in [%d, 10 min]", FB_WIN);
  const int F = (int)fbank_frames_of(n_samples);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t n_in = mem == MIA_MEM_HOST ? (size_t)n_samples : 0, n_fr = (size_t)F * FB_K, n_sp = (size_t)F * 2 * FB_NBP, n_fb = (size_t)F * FB_NMEL;
  float* ws = (float*)mia_workspace(ctx, (al(n_in) + al(n_fr) + al(n_sp) + al(n_fb)) * 4);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_in = ws; float* frames = d_in + al(n_in); float* spec = frames + al(n_fr); float* d_fb = spec + al(n_sp);
  const float* x = pcm;
  if (mem == MIA_MEM_HOST) { MIA_HIP(ctx, hipMemcpyAsync(d_in, pcm, (size_t)n_samples * 4, hipMemcpyHostToDevice, s)); x = d_in; }
  float* dst = mem == MIA_MEM_DEVICE ? fbank : d_fb;
  if (int rc = run_fbank(m, x, n_samples, F, frames, spec, dst, mean_norm != 0)) return rc;
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(fbank, d_fb, n_fb * 4, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}

namespace {

// a[h][t] = fb[t][h]: the fbank as the channels-last map [H = bin][W = frame][1] the first 2-D convolution reads
__global__ __launch_bounds__(256) void cam_transpose(const float* __restrict__ fb, float* __restrict__ a, int T) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)T * FB_NMEL) return;
  const int t = (int)(e / FB_NMEL), h = (int)(e % FB_NMEL);
  a[(int64_t)h * T + t] = fb[e];
}

size_t encoder_ws_floats(int T) {
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const int Tt = (T - 1) / 2 + 1, nseg = (Tt + SEG_LEN - 1) / SEG_LEN;
  return 3 * al((size_t)FB_NMEL * T * FCM_C) + al((size_t)T * 320) + al((size_t)Tt * 512) + 3 * al((size_t)Tt * 1024) + al((size_t)Tt * 128) +
         al((size_t)Tt * 32) + al((size_t)nseg * 32) + al((size_t)Tt * 512) + al(1024);
}

// mean-normalised fbank [T][80] (device) -> embedding [192] (device)
int run_encoder(mia_campplus* m, const float* fb, int T, float* ws, float* emb) {
  mia_ctx* ctx = m->ctx; hipStream_t s = ctx->stream;
  const int Tt = (T - 1) / 2 + 1, nseg = (Tt + SEG_LEN - 1) / SEG_LEN;       // TDNN: k5, stride 2, pad 2
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t n_map = al((size_t)FB_NMEL * T * FCM_C);
  float* A = ws; float* B = A + n_map; float* C = B + n_map;
  float* tin = C + n_map;                              // [T][320]
  float* cat[3];
  cat[0] = tin + al((size_t)T * 320);                  // [Tt][512]
  cat[1] = cat[0] + al((size_t)Tt * 512);              // [Tt][1024]
  cat[2] = cat[1] + al((size_t)Tt * 1024);             // [Tt][1024]
  float* tmp = cat[2] + al((size_t)Tt * 1024);         // [Tt][<= 1024]
  float* hb = tmp + al((size_t)Tt * 1024);             // [Tt][128]
  float* yb = hb + al((size_t)Tt * 128);               // [Tt][32]
  float* gate = yb + al((size_t)Tt * 32);              // [nseg][32]
  float* fin = gate + al((size_t)nseg * 32);           // [Tt][512]
  float* st = fin + al((size_t)Tt * 512);              // [1024]
  const int64_t map_t = FCM_C, map_h = (int64_t)T * FCM_C;                   // strides of a [H][T][32] map
  auto nblk = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  // ---- FCM (CAMPPlus.swift:246-324): bins 80 -> 40 -> 20 -> 10
  hipLaunchKernelGGL(cam_transpose, nblk((int64_t)T * FB_NMEL), dim3(256), 0, s, fb, A, T);
  run_conv2(s, m->conv1, A, 80, T, nullptr, B, 80, 1, 1, map_h, map_t, 1);
  // BasicResBlock (:180-242): relu(bn2(conv2(relu(bn1(conv1 x)))) + shortcut(x))
  run_conv2(s, m->res[0].c1, B, 80, T, nullptr, C, 40, 2, 1, map_h, map_t, 1);
  run_conv2(s, m->res[0].sc, B, 80, T, nullptr, A, 40, 2, 0, map_h, map_t, 1);
  run_conv2(s, m->res[0].c2, C, 40, T, A, B, 40, 1, 1, map_h, map_t, 1);
  run_conv2(s, m->res[1].c1, B, 40, T, nullptr, C, 40, 1, 1, map_h, map_t, 1);
  run_conv2(s, m->res[1].c2, C, 40, T, B, A, 40, 1, 1, map_h, map_t, 1);
  run_conv2(s, m->res[2].c1, A, 40, T, nullptr, C, 20, 2, 1, map_h, map_t, 1);
  run_conv2(s, m->res[2].sc, A, 40, T, nullptr, B, 20, 2, 0, map_h, map_t, 1);
  run_conv2(s, m->res[2].c2, C, 20, T, B, A, 20, 1, 1, map_h, map_t, 1);
  run_conv2(s, m->res[3].c1, A, 20, T, nullptr, C, 20, 1, 1, map_h, map_t, 1);
  run_conv2(s, m->res[3].c2, C, 20, T, A, B, 20, 1, 1, map_h, map_t, 1);
  // last FCM layer writes [T][c * 10 + h], the (B, C*H, W) reshape of :317-322 with frames as rows
  run_conv2(s, m->conv2, B, 20, T, nullptr, tin, 10, 2, 1, 1, 320, 10);
  if (hipGetLastError() != hipSuccess) return mia_fail(ctx, MIA_ERR_DEVICE, "campplus: FCM launch failed");
  // ---- TDNN (:345-393) + dense blocks (:507-609) + transits (:613-638)
  if (int rc = run_lin(ctx, m->tdnn, tin, 320, T, cat[0], 512, Tt, 2, 1, 2, 6)) return rc;
  const int64_t ld[3] = {512, 1024, 1024};
  int ch = 128;
  for (int b = 0; b < 3; ++b) {
    for (const DenseLayer& d : m->layers[b]) {
      hipLaunchKernelGGL(cam_bn_relu, nblk((int64_t)Tt * d.cin), dim3(256), 0, s, cat[b], ld[b], tmp, (int64_t)d.cin, Tt, d.cin, d.bn1.scale, d.bn1.shift);
      if (int rc = run_lin(ctx, d.lin1, tmp, d.cin, Tt, hb, 128, Tt, 1, 1, 0, 6)) return rc;
      if (int rc = run_lin(ctx, d.local, hb, 128, Tt, yb, 32, Tt, 1, d.dil, d.dil, 0)) return rc;
      hipLaunchKernelGGL(cam_gate, dim3(nseg), dim3(1024), 0, s, hb, Tt, d.w1, d.b1, d.w2, d.b2, gate);
      hipLaunchKernelGGL(cam_apply, nblk((int64_t)Tt * 32), dim3(256), 0, s, yb, gate, cat[b] + d.cin, ld[b], Tt);
    }
    ch += 32 * (int)m->layers[b].size();
    hipLaunchKernelGGL(cam_bn_relu, nblk((int64_t)Tt * ch), dim3(256), 0, s, cat[b], ld[b], tmp, (int64_t)ch, Tt, ch, m->tbn[b].scale, m->tbn[b].shift);
    float* dst = b < 2 ? cat[b + 1] : fin;
    if (int rc = run_lin(ctx, m->tlin[b], tmp, ch, Tt, dst, b < 2 ? ld[b + 1] : 512, Tt, 1, 1, 0, 0)) return rc;
    ch /= 2;
  }
  // ---- BN-ReLU -> statistics pooling -> dense (:765-781)
  hipLaunchKernelGGL(cam_stats, dim3((ch + 255) / 256), dim3(256), 0, s, fin, (int64_t)512, Tt, ch, m->out_bn.scale, m->out_bn.shift, st);
  hipLaunchKernelGGL(cam_dense, dim3(MIA_CAMPPLUS_DIM), dim3(64), 0, s, st, m->dense_w, m->dense_scale, m->dense_shift, emb, 2 * ch);
  return hipGetLastError() == hipSuccess ? MIA_OK : mia_fail(ctx, MIA_ERR_DEVICE, "campplus: launch failed");
}

}  // namespace

// CAMPPlus.callAsFunction on precomputed features: feats [T][80] (mean-normalised fbank) -> emb [192]
extern "C" int mia_campplus_forward(mia_campplus* m, const float* feats, int n_frames, float* emb, int mem) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, feats && emb && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "campplus_forward: bad argument");
  MIA_CHECK_ARG(ctx, n_frames >= 1 && n_frames <= 60000, "campplus_forward: n_frames must be in [1, 60000]");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t n_in = mem == MIA_MEM_HOST ? al((size_t)n_frames * FB_NMEL) : 0;
  float* ws = (float*)mia_workspace(ctx, (n_in + al(MIA_CAMPPLUS_DIM) + encoder_ws_floats(n_frames)) * 4);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_in = ws; float* d_emb = d_in + n_in; float* enc = d_emb + al(MIA_CAMPPLUS_DIM);
  const float* x = feats;
  if (mem == MIA_MEM_HOST) { MIA_HIP(ctx, hipMemcpyAsync(d_in, feats, (size_t)n_frames * FB_NMEL * 4, hipMemcpyHostToDevice, s)); x = d_in; }
  if (int rc = run_encoder(m, x, n_frames, enc, mem == MIA_MEM_DEVICE ? emb : d_emb)) return rc;
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(emb, d_emb, MIA_CAMPPLUS_DIM * 4, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}

// CAMPPlus.inference (CAMPPlus.swift:788-818) for one clip: pcm 16 kHz [n_samples] -> emb [192]
extern "C" int mia_campplus_embed(mia_campplus* m, const float* pcm, int64_t n_samples, float* emb, int mem) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, pcm && emb && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "campplus_embed: bad argument");
  MIA_CHECK_ARG(ctx, n_samples >= FB_WIN && n_samples <= (int64_t)16000 * 600, "campplus_embed: n_samples must be in [%d, 10 min]", FB_WIN);
  const int F = (int)fbank_frames_of(n_samples);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t n_in = mem == MIA_MEM_HOST ? al((size_t)n_samples) : 0, n_fr = al((size_t)F * FB_K), n_sp = al((size_t)F * 2 * FB_NBP), n_fb = al((size_t)F * FB_NMEL);
  float* ws = (float*)mia_workspace(ctx, (n_in + n_fr + n_sp + n_fb + al(MIA_CAMPPLUS_DIM) + encoder_ws_floats(F)) * 4);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_in = ws; float* frames = d_in + n_in; float* spec = frames + n_fr; float* fb = spec + n_sp; float* d_emb = fb + n_fb; float* enc = d_emb + al(MIA_CAMPPLUS_DIM);
  const float* x = pcm;
  if (mem == MIA_MEM_HOST) { MIA_HIP(ctx, hipMemcpyAsync(d_in, pcm, (size_t)n_samples * 4, hipMemcpyHostToDevice, s)); x = d_in; }
  if (int rc = run_fbank(m, x, n_samples, F, frames, spec, fb, true)) return rc;
  if (int rc = run_encoder(m, fb, F, enc, mem == MIA_MEM_DEVICE ? emb : d_emb)) return rc;
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(emb, d_emb, MIA_CAMPPLUS_DIM * 4, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}
