// mel_s3gen.hip -- 80-bin 24 kHz log-mel of the S3Gen / CosyVoice2 prompt features (SURVEY.md row 16 "mel", K1/K2 variant).
//
// Replaces s3genMelSpectrogram (Codec/S3Gen/Mel/S3GenMel.swift:43-102): reflect pad (n_fft - hop)/2 = 720, frames of 1920 at
// hop 480 (center: false), periodic Hann (hanningWindow(1921)[0..<1920]), |rfft|, slaney mel filterbank 0..8000 Hz
// (S3TokenizerUtils.swift:301-375), log(max(., 1e-5)).  Output [80][frames], channel-major like the reference.
//
// The 1920-point transform is ONE overlapping-window GEMM on the exact-fp32 matrix cores: the padded signal is viewed as rows of 32
// samples, frame f starts at row 15 f (x_row_mul = 15) and spans 60 taps, and the weight matrix is the Hann-windowed DFT basis for
// the bins the filterbank touches (k <= 640: 8 kHz), cosine rows then sine rows.  A small kernel takes magnitudes, applies the
// sparse filterbank rows and the log.  This runs once per prompt (CosyVoice2TTS.swift:370-430), not per utterance.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "codec.h"
#include "mia_device.h"
#include "mia_internal.h"

namespace {

constexpr int NFFT = 1920, HOP = 480, PAD = (NFFT - HOP) / 2, NMEL = 80, SR = 24000;
constexpr int NBIN = NFFT / 2 + 1;

struct S3GenMelTables { float* dft; float* fb_w; int* fb_meta; int nbp; int nnz; };

__global__ __launch_bounds__(256) void mel24_reflect_pad(const float* __restrict__ x, float* __restrict__ xp, int64_t T, int64_t Tp_alloc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= Tp_alloc) return;
  int64_t q = i - PAD;
  const int64_t Tp = T + 2 * PAD;
  float v = 0.f;
  if (i < Tp) {
    q = q < 0 ? -q : (q >= T ? 2 * (T - 1) - q : q);
    v = (q >= 0 && q < T) ? x[q] : 0.f;
  }
  xp[i] = v;
}

// spec [F][2 nbp] (cos part | sin part) -> out[m][f] = log(max(sum_k |X_k| w[m][k], 1e-5))
__global__ __launch_bounds__(256) void mel24_finish(const float* __restrict__ spec, const float* __restrict__ fb_w, const int* __restrict__ fb_meta,
                                                    float* __restrict__ out, int F, int nbp) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)F * NMEL) return;
  const int f = (int)(e / NMEL), m = (int)(e % NMEL);
  const int lo = fb_meta[m * 3], cnt = fb_meta[m * 3 + 1], off = fb_meta[m * 3 + 2];
  const float* re = spec + (int64_t)f * 2 * nbp + lo;
  const float* im = re + nbp;
  float acc = 0.f;
  for (int c = 0; c < cnt; ++c) acc = fmaf(sqrtf(re[c] * re[c] + im[c] * im[c]), fb_w[off + c], acc);
  out[(int64_t)m * F + f] = logf(fmaxf(acc, 1e-5f));
}

int get_tables(mia_ctx* ctx, S3GenMelTables** out) {
  if (ctx->s3gen_mel) { *out = (S3GenMelTables*)ctx->s3gen_mel; return MIA_OK; }
  // filterbank: fp32 scalar math like the reference (S3TokenizerUtils.swift:301-375), fMin 0, fMax 8000
  const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f, min_log_mel = min_log_hz / f_sp, logstep = logf(6.4f) / 27.0f;
  auto hz_to_mel = [&](float hz) { return hz >= min_log_hz ? min_log_mel + logf(hz / min_log_hz) / logstep : hz / f_sp; };
  auto mel_to_hz = [&](float mel) { return mel >= min_log_mel ? min_log_hz * expf(logstep * (mel - min_log_mel)) : f_sp * mel; };
  const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(8000.0f);
  std::vector<float> pts(NMEL + 2);
  for (int i = 0; i < NMEL + 2; ++i) pts[i] = mel_to_hz(mel_min + (float)i * (mel_max - mel_min) / (float)(NMEL + 1));
  std::vector<float> w; std::vector<int> meta(NMEL * 3);
  int kmax = 0;
  for (int m = 0; m < NMEL; ++m) {
    const float fl = pts[m], fc = pts[m + 1], fr = pts[m + 2], enorm = 2.0f / (pts[m + 2] - pts[m]);
    int lo = -1, hi = -1; std::vector<float> row(NBIN, 0.f);
    for (int k = 0; k < NBIN; ++k) {
      const float freq = (float)k * (float)SR / (float)NFFT;
      float v = 0.f;
      if (freq >= fl && freq <= fc) v = (freq - fl) / (fc - fl);
      else if (freq > fc && freq <= fr) v = (fr - freq) / (fr - fc);
      row[k] = v * enorm;
      if (row[k] != 0.f) { if (lo < 0) lo = k; hi = k; }
    }
    if (lo < 0) { lo = 0; hi = -1; }
    meta[m * 3] = lo; meta[m * 3 + 1] = hi - lo + 1; meta[m * 3 + 2] = (int)w.size();
    for (int k = lo; k <= hi; ++k) w.push_back(row[k]);
    if (hi > kmax) kmax = hi;
  }
  const int nbp = (kmax + 1 + 31) / 32 * 32;
  // Hann-windowed DFT basis rows: [2 nbp][1920]; window = hanningWindow(1921)[0..<1920] in fp32 (S3TokenizerUtils.swift:213-221)
  std::vector<float> dft((size_t)2 * nbp * NFFT, 0.f);
  const float factor = (float)M_PI / (float)NFFT;
  for (int k = 0; k <= kmax; ++k)
    for (int n = 0; n < NFFT; ++n) {
      const float win = 0.5f + 0.5f * cosf((float)(1 - (NFFT + 1) + 2 * n) * factor);
      const int r = (int)(((int64_t)k * n) % NFFT);
      const double ang = 2.0 * M_PI * (double)r / (double)NFFT;
      dft[(size_t)k * NFFT + n] = (float)(cos(ang) * (double)win);
      dft[(size_t)(nbp + k) * NFFT + n] = (float)(-sin(ang) * (double)win);
    }
  S3GenMelTables* t = (S3GenMelTables*)calloc(1, sizeof(S3GenMelTables));
  if (!t) return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "mel_s3gen: host allocation failed");
  t->nbp = nbp; t->nnz = (int)w.size();
  void* p[3] = {nullptr, nullptr, nullptr};
  const size_t bytes[3] = {dft.size() * 4, w.size() * 4, meta.size() * 4};
  const void* src[3] = {dft.data(), w.data(), meta.data()};
  for (int i = 0; i < 3; ++i) {
    if (hipMalloc(&p[i], bytes[i] + 64) != hipSuccess || hipMemcpyAsync(p[i], src[i], bytes[i], hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {     // the context's own stream, never the legacy stream (see logmel.hip)
      for (int j = 0; j <= i; ++j) if (p[j]) (void)hipFree(p[j]);
      free(t);
      return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "mel_s3gen: table upload failed");
    }
  }
  for (int i = 0; i < 3; ++i) ctx->table_allocs.push_back(p[i]);
  t->dft = (float*)p[0]; t->fb_w = (float*)p[1]; t->fb_meta = (int*)p[2];
  ctx->s3gen_mel = t;
  *out = t;
  return MIA_OK;
}

}  // namespace

extern "C" int64_t mia_mel_s3gen_frames(int64_t n_samples) {
  const int64_t Tp = n_samples + 2 * PAD;
  return Tp < NFFT ? 0 : 1 + (Tp - NFFT) / HOP;
}

extern "C" int mia_mel_s3gen(mia_ctx* ctx, const float* pcm, int64_t n_samples, float* mel, int mem) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, pcm && mel && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "mel_s3gen: bad argument");
  MIA_CHECK_ARG(ctx, n_samples > PAD && n_samples <= (int64_t)24000 * 600, "mel_s3gen: n_samples must be in (%d, 10 min]", PAD);
  const int64_t F = mia_mel_s3gen_frames(n_samples);
  MIA_CHECK_ARG(ctx, F > 0, "mel_s3gen: input too short for one frame");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  S3GenMelTables* tb = nullptr;
  if (int rc = get_tables(ctx, &tb)) return rc;
  hipStream_t s = ctx->stream;
  const int64_t Tp = n_samples + 2 * PAD;
  const int64_t rows = (Tp + 31) / 32 + 1;
  const size_t n_in = mem == MIA_MEM_HOST ? (size_t)n_samples : 0;
  const size_t n_xp = (size_t)rows * 32, n_spec = (size_t)F * 2 * tb->nbp, n_out = (size_t)F * NMEL;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  float* ws = (float*)mia_workspace(ctx, (al(n_in) + al(n_xp) + al(n_spec) + al(n_out)) * 4);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  float* d_in = ws; float* xp = d_in + al(n_in); float* spec = xp + al(n_xp); float* d_out = spec + al(n_spec);
  const float* x = pcm;
  if (mem == MIA_MEM_HOST) { MIA_HIP(ctx, hipMemcpyAsync(d_in, pcm, (size_t)n_samples * 4, hipMemcpyHostToDevice, s)); x = d_in; }
  hipLaunchKernelGGL(mel24_reflect_pad, dim3((unsigned)((n_xp + 255) / 256)), dim3(256), 0, s, x, xp, n_samples, (int64_t)n_xp);
  ConvGemmArgs g;
  g.X = xp; g.ldx = 32; g.T_in = (int)rows; g.W = tb->dft; g.Y = spec; g.ldy = 2 * tb->nbp; g.T_out = (int)F;
  g.M = (int)F; g.N = 2 * tb->nbp; g.Cin = 32; g.taps = NFFT / 32; g.x_row_mul = HOP / 32;
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "mel_s3gen: %s", e);
  if (codec_conv_gemm_launch(g, 1, s)) return mia_fail(ctx, MIA_ERR_DEVICE, "mel_s3gen: GEMM launch failed");
  float* dst = mem == MIA_MEM_DEVICE ? mel : d_out;
  hipLaunchKernelGGL(mel24_finish, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, spec, tb->fb_w, tb->fb_meta, dst, (int)F, tb->nbp);
  MIA_HIP(ctx, hipGetLastError());
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(mel, d_out, n_out * 4, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}

// ---- linear-interpolation resampler of the CosyVoice2 prompt path ---------------------------------------------------------------------------
// Replaces resampleAudio -> linearInterpolate1d (TTS/CosyVoice2/CosyVoice2TTS.swift:733-744, TTS/CosyVoice2/HiFiGAN/CosyHiFTGenerator.swift:17-60):
// new_T = int(float(T) * scale) (at least 1); src = (i + 0.5) * (float(T) / float(new_T)) - 0.5 clipped to [0, T - 1.001];
// out = x[floor] * (1 - frac) + x[min(floor + 1, T - 1)] * frac -- every operation in float32 and unfused, like the reference.
namespace {
__global__ __launch_bounds__(256) void resample_linear_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t T, int64_t newT, float ratio,
                                                              float clip_hi) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= newT) return;
  float idx = __fsub_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), ratio), 0.5f);
  idx = fminf(fmaxf(idx, 0.f), clip_hi);
  const float lo = floorf(idx);
  const int64_t ilo = (int64_t)lo, ihi = ilo + 1 < T - 1 ? ilo + 1 : T - 1;
  const float wh = __fsub_rn(idx, lo), wl = __fsub_rn(1.0f, wh);
  out[i] = __fadd_rn(__fmul_rn(x[ilo], wl), __fmul_rn(x[ihi], wh));
}
}  // namespace

extern "C" int64_t mia_resample_linear_len(int64_t n_samples, float scale) {
  const int64_t n = (int64_t)((float)n_samples * scale);
  return n == 0 ? 1 : n;
}

extern "C" int mia_resample_linear(mia_ctx* ctx, const float* x, int64_t n_samples, float scale, float* out, int mem) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, x && out && n_samples > 0 && n_samples < ((int64_t)1 << 24) && scale > 0.f && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE),
                "resample_linear: bad argument (n_samples must stay below 2^24: the reference indexes in float32)");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t newT = mia_resample_linear_len(n_samples, scale);
  hipStream_t s = ctx->stream;
  const float* d_x = x; float* d_o = out;
  if (mem == MIA_MEM_HOST) {
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };
    float* ws = (float*)mia_workspace(ctx, (al(n_samples) + al(newT)) * 4);
    if (!ws) return MIA_ERR_OUT_OF_MEMORY;
    MIA_HIP(ctx, hipMemcpyAsync(ws, x, (size_t)n_samples * 4, hipMemcpyHostToDevice, s));
    d_x = ws; d_o = ws + al(n_samples);
  }
  hipLaunchKernelGGL(resample_linear_kernel, dim3((unsigned)((newT + 255) / 256)), dim3(256), 0, s, d_x, d_o, n_samples, newT,
                     (float)n_samples / (float)newT, (float)n_samples - 1.001f);
  MIA_HIP(ctx, hipGetLastError());
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(out, d_o, (size_t)newT * 4, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}
