// logmel.hip -- fused STFT-power -> mel -> log10 front end for gfx950 (SURVEY.md table 2b rows K1+K2).
//
// Replaces whisperLogMelSpectrogram (STT/Whisper/WhisperAudio.swift:78-137) and the shared
// stft / reflectPad / melFilters helpers (Codec/S3Tokenizer/S3TokenizerUtils.swift:224-375).
//
// Pass 1 (logmel_pass1): one 256-thread workgroup per 32 consecutive frames of one clip.
//   * the 32 overlapping frames are gathered once from HBM (coalesced along the sample axis), multiplied
//     by the window and laid out frame-major in LDS with an odd row stride (401 floats) so the MFMA
//     A-operand column reads are bank-conflict free;
//   * the 400-point real DFT is a [32 x 400] x [400 x 2*224] contraction on the exact-f32 matrix cores
//     (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, so numerics equal an fp32 DFT);
//     cos and sin accumulators of one 32-bin tile share (lane, register), so |X|^2 is formed in-lane;
//   * power spectra go back to LDS, the (sparse, LDS-staged) Slaney filterbank is applied, log10 taken,
//     the fp32 result parked in a scratch tensor and the per-clip running max updated with one atomic per
//     workgroup.
// Pass 2 (logmel_pass2): clamp to (clip max - 8), (x+4)/4, cast, and scatter into the consumer's layout
//   (dense time-major, channel-major for S3, or the zero-row-padded layout the encoder's conv1 reads).
// Time-major outputs (the Whisper path) skip the fp32 round trip: pass 1 stores the FINAL value (x+4)/4 in the output type straight into
//   the consumer's layout, and only the clamp is left for afterwards.  Clamping commutes with the monotonic map and the rounding:
//   max(rnd((v+4)/4), rnd((c+4)/4)) == rnd((max(v, c)+4)/4), so logmel_clampfix rewrites -- in place, in the output type -- only the
//   32-frame blocks whose smallest value lies below the clip's floor (pass 1 records each block's minimum) plus the rows past the audio.
//   Traffic: audio once + output once (+ the few blocks that need the clamp) instead of + an fp32 mel tensor written and read back.
#include "mia_device.h"
#include "mia_internal.h"

#include <cmath>
#include <mutex>

namespace {

constexpr int NFFT = 400;
constexpr int HOP = 160;
constexpr int NBIN = 201;
constexpr int NBP = 224;      // bins padded to 7 tiles of 32
constexpr int FB = 32;        // frames per workgroup
constexpr int FSTR = 401;     // LDS row stride of the windowed frame matrix (odd: conflict-free columns)
constexpr int PSTR = 225;     // LDS row stride of the power matrix
constexpr int MAX_NNZ = 1024; // compact filterbank weights staged in LDS
constexpr int MAX_MELS = 128;

struct ClipInfo {
  int64_t off;   // first sample of the clip in pcm
  int64_t len;   // samples of audio
};

__device__ __forceinline__ float vsample(const float* __restrict__ x, int64_t L, int64_t Lp, int64_t v) {
  // virtual signal = audio || zeros(pad_right), reflect-padded (no edge repeat) by NFFT/2 on both sides
  if (v < 0) v = -v;
  if (v >= Lp) v = 2 * (Lp - 1) - v;
  return (v >= 0 && v < L) ? x[v] : 0.0f;
}

// out element (b, f, m) lives at out[b*clip_stride + (f+row_off)*row_stride + m*col_stride]
template <typename OutT>
__device__ __forceinline__ void store_out(void* out, size_t idx, float v);
template <>
__device__ __forceinline__ void store_out<float>(void* out, size_t idx, float v) { ((float*)out)[idx] = v; }
template <>
__device__ __forceinline__ void store_out<BF16>(void* out, size_t idx, float v) { ((uint16_t*)out)[idx] = BF16::from_f32(v); }
template <>
__device__ __forceinline__ void store_out<F16>(void* out, size_t idx, float v) { ((uint16_t*)out)[idx] = F16::from_f32(v); }

template <typename OutT>
__device__ __forceinline__ float load_out(const void* out, size_t idx);
template <>
__device__ __forceinline__ float load_out<float>(const void* out, size_t idx) { return ((const float*)out)[idx]; }
template <>
__device__ __forceinline__ float load_out<BF16>(const void* out, size_t idx) { return BF16::to_f32(((const uint16_t*)out)[idx]); }
template <>
__device__ __forceinline__ float load_out<F16>(const void* out, size_t idx) { return F16::to_f32(((const uint16_t*)out)[idx]); }

template <typename OutT>
__device__ __forceinline__ float round_out(float v) { return v; }
template <>
__device__ __forceinline__ float round_out<BF16>(float v) { return BF16::to_f32(BF16::from_f32(v)); }
template <>
__device__ __forceinline__ float round_out<F16>(float v) { return F16::to_f32(F16::from_f32(v)); }

// DIRECT: store (v + 4) / 4 as OutT at out[b*clip_stride + (f+row_off)*row_stride + m] and the block's minimum v in tmp[b*gridDim.x + block]
template <typename OutT, bool DIRECT>
__global__ __launch_bounds__(256) void logmel_pass1(const float* __restrict__ pcm, const ClipInfo* __restrict__ clips,
                                                    int64_t pad_right, int64_t n_out, int n_mels,
                                                    const float* __restrict__ window, const float* __restrict__ twiddle,
                                                    const float* __restrict__ fb_w, const int* __restrict__ fb_meta,
                                                    int fb_nnz, float* __restrict__ tmp, int* __restrict__ gmax,
                                                    void* __restrict__ out, int64_t clip_stride, int64_t row_stride, int64_t row_off) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* F = reinterpret_cast<float*>(smem_raw);            // [FB][FSTR]
  float* P = F + FB * FSTR;                                 // [FB][PSTR]
  float* W = P + FB * PSTR;                                 // [MAX_NNZ]
  int* META = reinterpret_cast<int*>(W + MAX_NNZ);          // [MAX_MELS][3]
  float* RED = reinterpret_cast<float*>(META + MAX_MELS * 3);  // [8]: per-wave max, per-wave min

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const ClipInfo ci = clips[b];
  const int64_t L = ci.len;
  const int64_t Lp = L + pad_right;
  const int64_t F_total = Lp / HOP;                          // frames of the padded utterance (last dropped)
  int64_t F_content = (L + NFFT / 2 + HOP - 1) / HOP;        // frames whose window touches audio
  if (F_content > F_total) F_content = F_total;
  const int64_t f0 = (int64_t)blockIdx.x * FB;
  if (f0 >= F_content) return;                               // uniform per workgroup
  const float* x = pcm + ci.off;

  // ---- stage filterbank + windowed frames --------------------------------------------------
  for (int i = tid; i < fb_nnz; i += 256) W[i] = fb_w[i];
  for (int i = tid; i < n_mels * 3; i += 256) META[i] = fb_meta[i];
  for (int idx = tid; idx < FB * NFFT; idx += 256) {
    const int i = idx / NFFT, k = idx - i * NFFT;
    const int64_t v = (f0 + i) * HOP + k - NFFT / 2;
    F[i * FSTR + k] = vsample(x, L, Lp, v) * window[k];
  }
  __syncthreads();

  // ---- DFT on the f32 matrix cores ------------------------------------------------------------
  // A[i][k] = F[i][k] (lane: i = lane&31, k = k0 + (lane>>5)); B[k][j] = twiddle[k][c][j]
  const int ai = (lane & 31) * FSTR + (lane >> 5);
  for (int t = wave; t < NBP / 32; t += 4) {
    f32x16 accC = {0}, accS = {0};
    const float* twc = twiddle + (size_t)(lane >> 5) * (2 * NBP) + 32 * t + (lane & 31);
#pragma unroll 8
    for (int k0 = 0; k0 < NFFT; k0 += 2) {
      const float a = F[ai + k0];
      const float bc = twc[(size_t)k0 * (2 * NBP)];
      const float bs = twc[(size_t)k0 * (2 * NBP) + NBP];
      accC = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bc, accC, 0, 0, 0);
      accS = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bs, accS, 0, 0, 0);
    }
    const int j = 32 * t + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      P[i * PSTR + j] = accC[r] * accC[r] + accS[r] * accS[r];   // pow(abs(X), 2)
    }
  }
  __syncthreads();

  // ---- mel filterbank (sparse rows, ascending bin order) + log10 --------------------------------
  float vmax = -INFINITY, vmin = INFINITY;
  for (int idx = tid; idx < FB * n_mels; idx += 256) {
    const int i = idx / n_mels, m = idx - i * n_mels;
    const int64_t f = f0 + i;
    if (f >= F_content) continue;
    const int lo = META[m * 3 + 0], cnt = META[m * 3 + 1], off = META[m * 3 + 2];
    const float* p = P + i * PSTR + lo;
    float acc = 0.0f;
    for (int c = 0; c < cnt; ++c) acc = fmaf(p[c], W[off + c], acc);
    const float v = log10f(fmaxf(acc, 1e-10f));
    vmax = fmaxf(vmax, v);
    if (f < n_out) {
      if (DIRECT) { vmin = fminf(vmin, v); store_out<OutT>(out, (size_t)(b * clip_stride + (f + row_off) * row_stride + m), (v + 4.0f) / 4.0f); }
      else tmp[((size_t)b * n_out + f) * n_mels + m] = v;
    }
  }
  vmax = wave_max(vmax);
  if (DIRECT) vmin = -wave_max(-vmin);
  if (lane == 0) { RED[wave] = vmax; if (DIRECT) RED[4 + wave] = vmin; }
  __syncthreads();
  if (tid == 0) {
    const float m4 = fmaxf(fmaxf(RED[0], RED[1]), fmaxf(RED[2], RED[3]));
    atomicMax(&gmax[b], float_to_ordered(m4));
    if (DIRECT) tmp[(size_t)b * gridDim.x + blockIdx.x] = fminf(fminf(RED[4], RED[5]), fminf(RED[6], RED[7]));
  }
}

// The clamp that is left after a DIRECT pass 1 (see the file header): block (x, b) owns output frames [32 x, 32 x + 32) of clip b.
template <typename OutT>
__global__ __launch_bounds__(256) void logmel_clampfix(const float* __restrict__ blkmin, int nblk1, const int* __restrict__ gmax,
                                                       const ClipInfo* __restrict__ clips, int64_t pad_right, int64_t n_out, int n_mels,
                                                       void* __restrict__ out, int64_t clip_stride, int64_t row_stride, int64_t row_off) {
  const int b = blockIdx.y;
  const int64_t L = clips[b].len;
  const int64_t F_total = (L + pad_right) / HOP;
  int64_t F_content = (L + NFFT / 2 + HOP - 1) / HOP;
  if (F_content > F_total) F_content = F_total;
  const int64_t f0 = (int64_t)blockIdx.x * FB;
  const float floor_v = ordered_to_float(gmax[b]) - 8.0f;
  const bool all_content = f0 + FB <= F_content && f0 + FB <= n_out;
  const bool need_clamp = f0 < F_content && blkmin[(size_t)b * nblk1 + blockIdx.x] < floor_v;
  if (all_content && !need_clamp) return;                                   // the common case: nothing to do for this block
  const float floor_o = (floor_v + 4.0f) / 4.0f;                            // the clamp in output units (rounded by store_out like any value)
  const float floor_r = round_out<OutT>(floor_o);                           // ... and as the output type holds it
  const float pad_o = (fmaxf(-10.0f, floor_v) + 4.0f) / 4.0f;               // frames whose window holds no audio: log10(1e-10) = -10
  for (int idx = threadIdx.x; idx < FB * n_mels; idx += 256) {
    const int i = idx / n_mels, m = idx - i * n_mels;
    const int64_t f = f0 + i;
    if (f >= n_out) continue;
    const size_t o = (size_t)(b * clip_stride + (f + row_off) * row_stride + m);
    if (f >= F_total) store_out<OutT>(out, o, 0.0f);                        // padOrTrimMel pads with 0.0 (WhisperSTT.swift:624-635)
    else if (f >= F_content) store_out<OutT>(out, o, pad_o);
    else if (need_clamp && load_out<OutT>(out, o) < floor_r) store_out<OutT>(out, o, floor_o);   // max in the OUTPUT type
  }
}

__global__ void logmel_init(int* gmax, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) gmax[i] = float_to_ordered(-10.0f);   // log10(1e-10): lower bound of every value
}

template <typename OutT, bool CHANNEL_MAJOR>
__global__ __launch_bounds__(256) void logmel_pass2(const float* __restrict__ tmp, const int* __restrict__ gmax,
                                                    const ClipInfo* __restrict__ clips, int64_t pad_right,
                                                    int64_t n_out, int n_mels, void* __restrict__ out,
                                                    int64_t clip_stride, int64_t row_stride, int64_t col_stride,
                                                    int64_t row_off) {
  const int b = blockIdx.y;
  const int64_t total = n_out * n_mels;
  const int64_t L = clips[b].len;
  const int64_t F_total = (L + pad_right) / HOP;
  int64_t F_content = (L + NFFT / 2 + HOP - 1) / HOP;
  if (F_content > F_total) F_content = F_total;
  const float floor_v = ordered_to_float(gmax[b]) - 8.0f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t f, m;
    if (CHANNEL_MAJOR) { m = e / n_out; f = e - m * n_out; }   // writes coalesced along frames
    else { f = e / n_mels; m = e - f * n_mels; }               // writes coalesced along mels
    float o;
    if (f >= F_total) {
      o = 0.0f;                                                 // padOrTrimMel pads with 0.0 (WhisperSTT.swift:624-635)
    } else {
      const float v = f < F_content ? tmp[((size_t)b * n_out + f) * n_mels + m] : -10.0f;
      o = (fmaxf(v, floor_v) + 4.0f) / 4.0f;
    }
    store_out<OutT>(out, (size_t)(b * clip_stride + (f + row_off) * row_stride + m * col_stride), o);
  }
}

// ---- host-side table construction (fp32 scalar math, mirrors S3TokenizerUtils.swift:301-375) ------
void build_mel_filters(int n_mels, std::vector<float>& dense) {
  const float f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
  const float min_log_mel = min_log_hz / f_sp;
  const float logstep = logf(6.4f) / 27.0f;
  auto hz_to_mel = [&](float hz) { return hz >= min_log_hz ? min_log_mel + logf(hz / min_log_hz) / logstep : hz / f_sp; };
  auto mel_to_hz = [&](float mel) { return mel >= min_log_mel ? min_log_hz * expf(logstep * (mel - min_log_mel)) : f_sp * mel; };
  const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(8000.0f);
  std::vector<float> pts(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) pts[i] = mel_to_hz(mel_min + (float)i * (mel_max - mel_min) / (float)(n_mels + 1));
  dense.assign((size_t)n_mels * NBIN, 0.0f);
  for (int m = 0; m < n_mels; ++m) {
    const float fl = pts[m], fc = pts[m + 1], fr = pts[m + 2];
    const float enorm = 2.0f / (pts[m + 2] - pts[m]);
    for (int k = 0; k < NBIN; ++k) {
      const float freq = (float)k * 16000.0f / (float)NFFT;
      float w = 0.0f;
      if (freq >= fl && freq <= fc) w = (freq - fl) / (fc - fl);
      else if (freq > fc && freq <= fr) w = (fr - freq) / (fr - fc);
      dense[(size_t)m * NBIN + k] = w * enorm;
    }
  }
}

// window_kind 0: Whisper symmetric Hann (WhisperAudio.swift:32-44); 1: periodic Hann (S3TokenizerUtils.swift:117,213-221)
int get_tables(mia_ctx* ctx, int n_mels, int window_kind, mia_ctx::MelTables** out) {
  for (auto& t : ctx->mel_tables)
    if (t.n_mels == n_mels && t.window_kind == window_kind) { *out = &t; return MIA_OK; }
  mia_ctx::MelTables t;
  t.n_mels = n_mels;
  t.window_kind = window_kind;
  std::vector<float> win(NFFT);
  if (window_kind == 0) {
    const float factor = 2.0f * (float)M_PI / (float)(NFFT - 1);
    for (int n = 0; n < NFFT; ++n) win[n] = 0.5f * (1.0f - cosf((float)n * factor));
  } else {
    const int Lw = NFFT + 1;
    const float factor = (float)M_PI / (float)(Lw - 1);
    for (int n = 0; n < NFFT; ++n) win[n] = 0.5f + 0.5f * cosf((float)(1 - Lw + 2 * n) * factor);
  }
  std::vector<float> tw((size_t)NFFT * 2 * NBP, 0.0f);
  for (int k = 0; k < NFFT; ++k)
    for (int j = 0; j < NBIN; ++j) {
      const int r = (int)(((int64_t)k * j) % NFFT);            // exact argument reduction
      const double ang = 2.0 * M_PI * (double)r / (double)NFFT;
      tw[((size_t)k * 2 + 0) * NBP + j] = (float)cos(ang);
      tw[((size_t)k * 2 + 1) * NBP + j] = (float)sin(ang);
    }
  std::vector<float> dense;
  build_mel_filters(n_mels, dense);
  std::vector<float> w;
  std::vector<int> meta((size_t)n_mels * 3);
  for (int m = 0; m < n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < NBIN; ++k)
      if (dense[(size_t)m * NBIN + k] != 0.0f) { if (lo < 0) lo = k; hi = k; }
    if (lo < 0) { lo = 0; hi = -1; }
    meta[m * 3 + 0] = lo;
    meta[m * 3 + 1] = hi - lo + 1;
    meta[m * 3 + 2] = (int)w.size();
    for (int k = lo; k <= hi; ++k) w.push_back(dense[(size_t)m * NBIN + k]);
  }
  if ((int)w.size() > MAX_NNZ) return mia_fail(ctx, MIA_ERR_UNSUPPORTED, "filterbank has %zu non-zeros (> %d)", w.size(), MAX_NNZ);
  if (w.empty()) w.push_back(0.0f);
  t.fb_nnz = (int)w.size();
  MIA_HIP(ctx, hipMalloc(&t.window, NFFT * sizeof(float)));
  MIA_HIP(ctx, hipMalloc(&t.twiddle, tw.size() * sizeof(float)));
  MIA_HIP(ctx, hipMalloc(&t.fb_w, w.size() * sizeof(float)));
  MIA_HIP(ctx, hipMalloc(&t.fb_meta, meta.size() * sizeof(int)));
  // uploads go through the context's OWN stream: a synchronous hipMemcpy runs on the legacy stream, which is illegal while another host
  // thread captures a step graph on a blocking stream (bench.py's replicas; seen as "operation would make the legacy stream depend on a
  // capturing blocking stream" in a 2-rank rehearsal)
  MIA_HIP(ctx, hipMemcpyAsync(t.window, win.data(), NFFT * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MIA_HIP(ctx, hipMemcpyAsync(t.twiddle, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MIA_HIP(ctx, hipMemcpyAsync(t.fb_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  MIA_HIP(ctx, hipMemcpyAsync(t.fb_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->mel_tables.push_back(t);
  *out = &ctx->mel_tables.back();
  return MIA_OK;
}

constexpr size_t PASS1_LDS = (size_t)(FB * FSTR + FB * PSTR + MAX_NNZ) * 4 + MAX_MELS * 3 * 4 + 32;   // ... + RED[8]

}  // namespace

// Internal entry used by the C ABI wrappers and by the fused Whisper pipeline.
//   pcm_dev        device pointer to the concatenated clips
//   out_dev        device pointer; element (b,f,m) at b*clip_stride + (f+row_off)*row_stride + m*col_stride
//   scratch        device scratch of at least logmel_scratch_bytes(B, n_out, n_mels) bytes
size_t mia_logmel_scratch_bytes(int B, int64_t n_out, int n_mels) {
  return align_up((size_t)B * sizeof(ClipInfo), 256) + align_up((size_t)B * sizeof(int), 256) +
         align_up((size_t)B * n_out * n_mels * sizeof(float), 256);
}

int mia_logmel_device(mia_ctx* ctx, const float* pcm_dev, const int64_t* offs_host, int B, int n_mels, int window_kind,
                      int64_t pad_right, int64_t n_out, void* out_dev, int out_dtype, bool channel_major,
                      int64_t clip_stride, int64_t row_stride, int64_t col_stride, int64_t row_off, void* scratch) {
  MIA_CHECK_ARG(ctx, B > 0, "logmel: B must be > 0 (got %d)", B);
  MIA_CHECK_ARG(ctx, n_mels > 0 && n_mels <= MAX_MELS, "logmel: n_mels must be in 1..%d (got %d)", MAX_MELS, n_mels);
  MIA_CHECK_ARG(ctx, n_out > 0 && pad_right >= 0, "logmel: n_frames_out must be > 0 and pad_right >= 0");
  MIA_CHECK_ARG(ctx, out_dtype == MIA_F32 || out_dtype == MIA_F16 || out_dtype == MIA_BF16, "logmel: bad out_dtype %d", out_dtype);
  std::vector<ClipInfo> clips(B);
  int64_t max_content = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t len = offs_host[b + 1] - offs_host[b];
    if (len <= 0 || len + pad_right <= NFFT / 2)
      return mia_fail(ctx, MIA_ERR_INVALID_AUDIO, "logmel: clip %d has %lld samples (+%lld pad): too short for STFT", b,
                      (long long)len, (long long)pad_right);
    clips[b].off = offs_host[b];
    clips[b].len = len;
    int64_t fc = (len + NFFT / 2 + HOP - 1) / HOP;
    const int64_t ft = (len + pad_right) / HOP;
    if (fc > ft) fc = ft;
    if (fc > max_content) max_content = fc;
  }
  mia_ctx::MelTables* tb = nullptr;
  int rc = get_tables(ctx, n_mels, window_kind, &tb);
  if (rc != MIA_OK) return rc;

  char* s = (char*)scratch;
  ClipInfo* d_clips = (ClipInfo*)s;
  s += align_up((size_t)B * sizeof(ClipInfo), 256);
  int* d_gmax = (int*)s;
  s += align_up((size_t)B * sizeof(int), 256);
  float* d_tmp = (float*)s;

  MIA_HIP(ctx, hipMemcpyAsync(d_clips, clips.data(), (size_t)B * sizeof(ClipInfo), hipMemcpyHostToDevice, ctx->stream));
  double alg_bytes = 0.0;   // algorithmic traffic: audio read once + output written once
  for (int b = 0; b < B; ++b) alg_bytes += (double)clips[b].len * 4.0 + (double)n_out * n_mels * mia_dtype_size(out_dtype);
  const int prof_rec = mia_prof_begin(ctx, MIA_PROF_LOGMEL, alg_bytes);
  hipLaunchKernelGGL(logmel_init, dim3((B + 255) / 256), dim3(256), 0, ctx->stream, d_gmax, B);
  // 86 KB of dynamic LDS: above the 64 KB default cap.  Several contexts (bench.py's replica threads) reach this concurrently.
  static std::once_flag lds_attr_once;
  static hipError_t lds_attr_rc = hipSuccess;
  std::call_once(lds_attr_once, [] {
    const void* fns[4] = {(const void*)logmel_pass1<float, false>, (const void*)logmel_pass1<float, true>, (const void*)logmel_pass1<F16, true>,
                          (const void*)logmel_pass1<BF16, true>};
    for (const void* f : fns) { const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PASS1_LDS); if (e != hipSuccess) lds_attr_rc = e; }
  });
  MIA_HIP(ctx, lds_attr_rc);
  const unsigned nblk1 = (unsigned)((max_content + FB - 1) / FB);
  // time-major outputs with unit column stride take the single-pass form (file header); the channel-major (S3) layout keeps the fp32
  // intermediate: its transposed stores are coalesced only along frames, which pass 2 walks
  const bool direct = !channel_major && col_stride == 1;
#define LAUNCH1(T, D)                                                                                                        \
  hipLaunchKernelGGL((logmel_pass1<T, D>), dim3(nblk1, (unsigned)B), dim3(256), PASS1_LDS, ctx->stream, pcm_dev, d_clips, pad_right, n_out, \
                     n_mels, tb->window, tb->twiddle, tb->fb_w, tb->fb_meta, tb->fb_nnz, d_tmp, d_gmax, out_dev, clip_stride, row_stride, row_off)
  if (max_content > 0) {
    if (!direct) LAUNCH1(float, false);
    else if (out_dtype == MIA_F32) LAUNCH1(float, true);
    else if (out_dtype == MIA_F16) LAUNCH1(F16, true);
    else LAUNCH1(BF16, true);
  }
#undef LAUNCH1
  if (direct) {
    const dim3 gridf((unsigned)((n_out + FB - 1) / FB), (unsigned)B);
#define LAUNCHF(T) hipLaunchKernelGGL((logmel_clampfix<T>), gridf, dim3(256), 0, ctx->stream, d_tmp, (int)nblk1, d_gmax, d_clips, pad_right, n_out, n_mels, \
                                      out_dev, clip_stride, row_stride, row_off)
    if (out_dtype == MIA_F32) LAUNCHF(float); else if (out_dtype == MIA_F16) LAUNCHF(F16); else LAUNCHF(BF16);
#undef LAUNCHF
  } else {
    const int64_t total = n_out * n_mels;
    unsigned gx = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
    dim3 grid2(gx, (unsigned)B);
#define LAUNCH2(T, CM)                                                                                               \
    hipLaunchKernelGGL((logmel_pass2<T, CM>), grid2, dim3(256), 0, ctx->stream, d_tmp, d_gmax, d_clips, pad_right, n_out, \
                       n_mels, out_dev, clip_stride, row_stride, col_stride, row_off)
    if (channel_major) {
      if (out_dtype == MIA_F32) LAUNCH2(float, true);
      else if (out_dtype == MIA_F16) LAUNCH2(F16, true);
      else LAUNCH2(BF16, true);
    } else {
      if (out_dtype == MIA_F32) LAUNCH2(float, false);
      else if (out_dtype == MIA_F16) LAUNCH2(F16, false);
      else LAUNCH2(BF16, false);
    }
#undef LAUNCH2
  }
  mia_prof_end(ctx, prof_rec);
  MIA_HIP(ctx, hipGetLastError());
  // clips[] lives on the host stack of this call: the async H2D copy above must have consumed it.
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIA_OK;
}

static int logmel_public(mia_ctx* ctx, const float* pcm, const int64_t* offs, int B, int n_mels, int64_t pad_right,
                         int64_t n_out, void* mel, int out_dtype, int mem, int window_kind, bool channel_major) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, pcm && offs && mel, "logmel: null pointer argument");
  MIA_CHECK_ARG(ctx, B > 0 && n_out > 0, "logmel: B and n_frames_out must be > 0");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "logmel: bad mem space %d", mem);
  MIA_CHECK_ARG(ctx, out_dtype == MIA_F32 || out_dtype == MIA_F16 || out_dtype == MIA_BF16, "logmel: bad out_dtype %d", out_dtype);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t total_samples = offs[B] - offs[0];
  MIA_CHECK_ARG(ctx, total_samples > 0, "logmel: empty input");
  const size_t sc = mia_logmel_scratch_bytes(B, n_out, n_mels);
  const size_t out_bytes = (size_t)B * n_out * n_mels * mia_dtype_size(out_dtype);
  const size_t pcm_bytes = (size_t)(offs[B]) * sizeof(float);
  size_t need = sc;
  if (mem == MIA_MEM_HOST) need += align_up(pcm_bytes, 256) + align_up(out_bytes, 256);
  char* ws = (char*)mia_workspace(ctx, need);
  if (!ws) return MIA_ERR_OUT_OF_MEMORY;
  const float* d_pcm = pcm;
  void* d_out = mel;
  if (mem == MIA_MEM_HOST) {
    float* p = (float*)(ws + sc);
    d_out = ws + sc + align_up(pcm_bytes, 256);
    MIA_HIP(ctx, hipMemcpyAsync(p, pcm, pcm_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_pcm = p;
  }
  int64_t clip_stride, row_stride, col_stride;
  if (channel_major) { clip_stride = n_out * n_mels; row_stride = 1; col_stride = n_out; }
  else { clip_stride = n_out * n_mels; row_stride = n_mels; col_stride = 1; }
  int rc = mia_logmel_device(ctx, d_pcm, offs, B, n_mels, window_kind, pad_right, n_out, d_out, out_dtype, channel_major,
                             clip_stride, row_stride, col_stride, 0, ws);
  if (rc != MIA_OK) return rc;
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(mel, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MIA_OK;
}

extern "C" int mia_logmel_whisper(mia_ctx* ctx, const float* pcm, const int64_t* offs, int B, int n_mels, int64_t pad_right,
                                  int64_t n_frames_out, void* mel, int out_dtype, int mem) {
  return logmel_public(ctx, pcm, offs, B, n_mels, pad_right, n_frames_out, mel, out_dtype, mem, 0, false);
}

extern "C" int mia_logmel_s3(mia_ctx* ctx, const float* pcm, const int64_t* offs, int B, int n_mels, int64_t pad_right,
                             int64_t n_frames_out, void* mel, int out_dtype, int mem) {
  return logmel_public(ctx, pcm, offs, B, n_mels, pad_right, n_frames_out, mel, out_dtype, mem, 1, true);
}
