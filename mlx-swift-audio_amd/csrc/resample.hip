// resample.hip -- anti-aliased sample-rate conversion of input audio on the device (first half).
//
// The reference converts with AVAudioConverter (Audio/AudioResampler.swift:15-88; STT/Whisper/WhisperEngine.swift:328-369 loads and
// mixes to mono first).  Apple's converter is closed and its filter is not specified in the reference's sources, so there is nothing
// to restate; this is a DOCUMENTED band-limited interpolator instead, not a parity claim:
//   rational ratio L / M = to / from (reduced); y[j] = sum_t h_p[t] x[i0 - HW + t],  i0 = floor(j M / L),  p = (j M) mod L,
//   h_p[t] = 2 fc sinc(2 fc d) * kaiser_beta(d / HW'),  d = (i0 - HW + t) - j M / L,  fc = 0.5 * rolloff * min(1, L / M),
//   HW' = zeros / (2 fc) input samples (zeros = 16 zero crossings per side, rolloff 0.945, beta 8.6: about -90 dB stop band);
//   every phase is normalised to unit DC gain; n_out = floor(n L / M); samples outside [0, n) read as zero.
// The L phase filters are built on the host in float64 (a few thousand floats) and cached per ratio; the kernel is one coalesced
// read of x and one write of y (HBM-bound: 4 + 4 L/M bytes per input sample).
#include <cmath>
#include <numeric>

#include "mia_device.h"
#include "mia_internal.h"

namespace {

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ table,
                                                            int64_t n_in, int64_t n_out, int L, int M, int taps, int hw) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const int64_t jm = j * (int64_t)M;
  const int64_t i0 = jm / L;
  const int p = (int)(jm - i0 * L);
  const float* h = table + (int64_t)p * taps;
  float acc = 0.f;
  for (int t = 0; t < taps; ++t) {
    const int64_t i = i0 - hw + t;
    if (i >= 0 && i < n_in) acc += h[t] * x[i];
  }
  y[j] = acc;
}

double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  for (int k = 1; k < 64; ++k) { term *= (x / (2.0 * k)) * (x / (2.0 * k)); sum += term; if (term < 1e-18 * sum) break; }
  return sum;
}

struct SincTable { int from, to, L, M, taps, hw; float* dev; };

}  // namespace

struct mia_resampler_cache { std::vector<SincTable> tables; };

static const double kZeros = 16.0, kRolloff = 0.945, kBeta = 8.6;

extern "C" int64_t mia_resample_sinc_len(int64_t n_samples, int from_rate, int to_rate) {
  if (n_samples <= 0 || from_rate <= 0 || to_rate <= 0) return 0;
  const int g = std::gcd(from_rate, to_rate);
  return n_samples * (to_rate / g) / (from_rate / g);
}

// Host form of the L phase filters (float32 [L][taps]); also what tests/test_resample.py checks the device path against.
extern "C" int mia_resample_sinc_table(int from_rate, int to_rate, float* table, int capacity, int* L_out, int* taps_out) {
  if (from_rate <= 0 || to_rate <= 0) return MIA_ERR_INVALID_ARGUMENT;
  const int g = std::gcd(from_rate, to_rate);
  const int L = to_rate / g, M = from_rate / g;
  const double fc = 0.5 * kRolloff * std::min(1.0, (double)L / (double)M);
  const double hwd = kZeros / (2.0 * fc);
  const int hw = (int)std::ceil(hwd), taps = 2 * hw + 1;
  if (L_out) *L_out = L;
  if (taps_out) *taps_out = taps;
  if (!table) return MIA_OK;
  if ((int64_t)L * taps > capacity) return MIA_ERR_INVALID_ARGUMENT;
  const double i0b = bessel_i0(kBeta), pi = 3.14159265358979323846;
  for (int p = 0; p < L; ++p) {
    std::vector<double> h(taps);
    double sum = 0.0;
    for (int t = 0; t < taps; ++t) {
      const double d = (double)(t - hw) - (double)p / (double)L;      // input index minus the output's position
      const double r = d / hwd;
      double w = 0.0;
      if (std::fabs(r) < 1.0) {
        const double a = 2.0 * fc * d * pi;
        const double sinc = std::fabs(a) < 1e-12 ? 1.0 : std::sin(a) / a;
        w = 2.0 * fc * sinc * bessel_i0(kBeta * std::sqrt(1.0 - r * r)) / i0b;
      }
      h[t] = w; sum += w;
    }
    for (int t = 0; t < taps; ++t) table[(int64_t)p * taps + t] = (float)(h[t] / sum);
  }
  return MIA_OK;
}

extern "C" int mia_resample_sinc(mia_ctx* ctx, const float* x, int64_t n_samples, int from_rate, int to_rate, float* out, int64_t out_capacity,
                                 int64_t* n_out, int mem) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, x && out && n_samples > 0 && from_rate > 0 && to_rate > 0 && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "resample_sinc: bad argument");
  const int64_t N = mia_resample_sinc_len(n_samples, from_rate, to_rate);
  MIA_CHECK_ARG(ctx, N > 0 && out_capacity >= N, "resample_sinc: output buffer too small (%lld < %lld)", (long long)out_capacity, (long long)N);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (!ctx->resampler) ctx->resampler = new mia_resampler_cache();
  SincTable* tb = nullptr;
  for (SincTable& t : ctx->resampler->tables) if (t.from == from_rate && t.to == to_rate) tb = &t;
  if (!tb) {
    int L = 0, taps = 0;
    mia_resample_sinc_table(from_rate, to_rate, nullptr, 0, &L, &taps);
    MIA_CHECK_ARG(ctx, (int64_t)L * taps <= (1 << 24), "resample_sinc: ratio %d/%d needs too many filter phases", to_rate, from_rate);
    std::vector<float> host((size_t)L * taps);
    mia_resample_sinc_table(from_rate, to_rate, host.data(), (int)host.size(), nullptr, nullptr);
    float* dev = nullptr;
    MIA_HIP(ctx, hipMalloc((void**)&dev, host.size() * 4));
    ctx->table_allocs.push_back(dev);
    MIA_HIP(ctx, hipMemcpyAsync(dev, host.data(), host.size() * 4, hipMemcpyHostToDevice, s));   // own stream, never the legacy stream (see logmel.hip)
    MIA_HIP(ctx, hipStreamSynchronize(s));
    const int g = std::gcd(from_rate, to_rate);
    ctx->resampler->tables.push_back(SincTable{from_rate, to_rate, to_rate / g, from_rate / g, taps, (taps - 1) / 2, dev});
    tb = &ctx->resampler->tables.back();
  }
  const float* d_x = x; float* d_o = out;
  if (mem == MIA_MEM_HOST) {
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };
    float* ws = (float*)mia_workspace(ctx, (al(n_samples) + al(N)) * 4);
    if (!ws) return MIA_ERR_OUT_OF_MEMORY;
    MIA_HIP(ctx, hipMemcpyAsync(ws, x, (size_t)n_samples * 4, hipMemcpyHostToDevice, s));
    d_x = ws; d_o = ws + al(n_samples);
  }
  if (tb->L == 1 && tb->M == 1) MIA_HIP(ctx, hipMemcpyAsync(d_o, d_x, (size_t)N * 4, hipMemcpyDeviceToDevice, s));     // same rate: the reference returns its input
  else hipLaunchKernelGGL(resample_sinc_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, d_x, d_o, tb->dev, n_samples, N, tb->L, tb->M, tb->taps, tb->hw);
  MIA_HIP(ctx, hipGetLastError());
  if (n_out) *n_out = N;
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(out, d_o, (size_t)N * 4, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}

void mia_resampler_free(mia_ctx* ctx) {
  delete ctx->resampler;
  ctx->resampler = nullptr;
}
