// s3tok.hip -- S3TokenizerV2 / V3 speech tokenizer (audio log-mel -> 25 Hz token ids in [0, 6561)) on gfx950, fp32 like the
// reference.
//
// Replaces S3TokenizerV2.quantize -> AudioEncoderV2 -> FSMNMultiHeadAttention -> FSQCodebook.encode
// (Codec/S3Tokenizer/S3Tokenizer.swift:474-494, 396-436, 225-315, 149-168).  Each clip is processed at its exact length, which
// equals the reference's padded-batch + mask arithmetic on every valid position (masks only zero padded frames / keys).
//   conv1, conv2 (k3, stride 2, GELU), q|k|v, out, MLP: tap-structured exact-fp32 MFMA GEMM of codec_kernels.hip
//   RoPE (Swift port's table: freqs[j] = theta^(-j/64), j < 32; halves duplicated; rotation [-x_R, x_L]) + d^-1/4 scaling
//   FSMN memory: depthwise k31 conv over V, zero padded (15,15), + V, added after the output projection
//   attention: fp32, LDS-tiled (32 queries x 64-key tiles), exact softmax
//   FSQ: round(tanh(W x + b) * 0.999) + 1 in base 3 (rintf = half-to-even like MLX round)
#include <cmath>
#include <mutex>
#include <map>
#include <string>
#include <vector>

#include "codec.h"
#include "mia_device.h"
#include "mia_internal.h"
#include "ops.h"

namespace {

// mel [n_mels][T] (channel-major, as the reference passes it) -> time-major rows [T+2][n_mels] with zero rows at both ends
__global__ void s3_transpose_pad(const float* __restrict__ mel, int64_t ld_mel, float* __restrict__ out, int T, int n_mels) {
  const int64_t total = (int64_t)(T + 2) * n_mels;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n_mels), m = (int)(e - (int64_t)r * n_mels);
    out[e] = (r == 0 || r == T + 1) ? 0.f : mel[(int64_t)m * ld_mel + (r - 1)];
  }
}

// in place on qkv [T][3D]: q and k <- (x*cos + rot(x)*sin) * scale, rot = [-x_R, x_L] per 64-wide head
__global__ void s3_rope_scale(float* __restrict__ qkv, int T, int D, float theta, float scale) {
  const int64_t total = (int64_t)T * 2 * (D / 2);          // one thread per (t, q|k, pair)
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e / D), rem = (int)(e - (int64_t)t * D);       // rem in [0, D): D/2 pairs for q then D/2 for k
    const int sec = rem / (D / 2), pr = rem - sec * (D / 2);
    const int head = pr / 32, j = pr - head * 32;
    float* x = qkv + (int64_t)t * 3 * D + sec * D + head * 64;
    const float ang = (float)t * powf(theta, -(float)j / 64.0f);
    float sn, cs;
    sincosf(ang, &sn, &cs);
    const float xl = x[j], xr = x[j + 32];
    x[j] = (xl * cs - xr * sn) * scale;
    x[j + 32] = (xr * cs + xl * sn) * scale;
  }
}

// x[t][c] += v[t][c] + sum_k w[k][c] * v[t + k - K/2][c]      (FSMN memory, zero padded)
__global__ void s3_fsmn_add(float* __restrict__ x, const float* __restrict__ v, int64_t ldv, const float* __restrict__ w, int T, int C, int K) {
  const int c4n = C >> 2;
  const int64_t total = (int64_t)T * c4n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int t = (int)(e / c4n), c = (int)(e - (int64_t)t * c4n) * 4;
    float4 acc = *reinterpret_cast<const float4*>(v + (int64_t)t * ldv + c);
    for (int k = 0; k < K; ++k) {
      const int ts = t + k - K / 2;
      if (ts < 0 || ts >= T) continue;
      const float4 vv = *reinterpret_cast<const float4*>(v + (int64_t)ts * ldv + c);
      const float4 wk = *reinterpret_cast<const float4*>(w + (int64_t)k * C + c);
      acc.x = fmaf(wk.x, vv.x, acc.x); acc.y = fmaf(wk.y, vv.y, acc.y); acc.z = fmaf(wk.z, vv.z, acc.z); acc.w = fmaf(wk.w, vv.w, acc.w);
    }
    float4* xp = reinterpret_cast<float4*>(x + (int64_t)t * C + c);
    float4 xv = *xp;
    xv.x += acc.x; xv.y += acc.y; xv.z += acc.z; xv.w += acc.w;
    *xp = xv;
  }
}

// fp32 attention, head dim 64: block = (head, 32 queries), 256 threads; scores of the 32 queries against all keys in LDS
__global__ __launch_bounds__(256) void s3_attention(const float* __restrict__ qkv, float* __restrict__ out, int T, int D) {
  extern __shared__ float sm[];
  const int Tp = (T + 63) & ~63;
  float* S = sm;                       // [32][Tp + 1]
  float* Qs = S + 32 * (Tp + 1);       // [32][65]
  float* KV = Qs + 32 * 65;            // [64][65]
  const int tid = threadIdx.x, h = blockIdx.y, q0 = blockIdx.x * 32;
  const int qi = tid >> 3, sub = tid & 7;                    // 8 threads per query row
  for (int e = tid; e < 32 * 64; e += 256) {
    const int r = e >> 6, d = e & 63;
    const int q = q0 + r;
    Qs[r * 65 + d] = q < T ? qkv[(int64_t)q * 3 * D + h * 64 + d] : 0.f;
  }
  for (int k0 = 0; k0 < T; k0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, d = e & 63;
      const int k = k0 + r;
      KV[r * 65 + d] = k < T ? qkv[(int64_t)k * 3 * D + D + h * 64 + d] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = sub + 8 * j;
      float acc = 0.f;
#pragma unroll 16
      for (int d = 0; d < 64; ++d) acc = fmaf(Qs[qi * 65 + d], KV[kk * 65 + d], acc);
      S[qi * (Tp + 1) + k0 + kk] = (k0 + kk < T) ? acc : -INFINITY;
    }
  }
  __syncthreads();
  // softmax of row qi by its 8 threads
  float m = -INFINITY;
  for (int k = sub; k < T; k += 8) m = fmaxf(m, S[qi * (Tp + 1) + k]);
  m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
  float sum = 0.f;
  for (int k = sub; k < Tp; k += 8) { const float p = k < T ? expf(S[qi * (Tp + 1) + k] - m) : 0.f; S[qi * (Tp + 1) + k] = p; sum += p; }
  sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k0 = 0; k0 < T; k0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, d = e & 63;
      const int k = k0 + r;
      KV[r * 65 + d] = k < T ? qkv[(int64_t)k * 3 * D + 2 * D + h * 64 + d] : 0.f;
    }
    __syncthreads();
    for (int kk = 0; kk < 64; ++kk) {
      const float p = S[qi * (Tp + 1) + k0 + kk];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(p, KV[kk * 65 + sub * 8 + j], acc[j]);
    }
  }
  const int q = q0 + qi;
  if (q < T) {
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(int64_t)q * D + h * 64 + sub * 8 + j] = acc[j] * inv;
  }
}

// FSQ: id = sum_i (round(tanh(w_i . x + b_i) * 0.999) + 1) * 3^i      one wave per token
__global__ __launch_bounds__(256) void s3_fsq(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, int32_t* __restrict__ ids, int T, int D) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T) return;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float xv = x[(int64_t)t * D + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(w[(int64_t)i * D + c], xv, acc[i]);
  }
  float mu = 0.f, pw = 1.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = wave_sum(acc[i]) + b[i];
    const float hq = rintf(tanhf(s) * 0.9990000128746033f) + 1.0f;     // S3Tokenizer.swift:156-159
    mu += hq * pw;
    pw *= 3.0f;
  }
  if (lane == 0) ids[t] = (int32_t)mu;
}

}  // namespace

struct S3Block {
  float *attn_ln_g, *attn_ln_b, *mlp_ln_g, *mlp_ln_b;
  float *wqkv, *bqkv, *wo, *bo, *fsmn, *w1, *b1, *w2, *b2;
};

struct mia_s3tok {
  mia_ctx* ctx = nullptr;
  mia_s3_config cfg{};
  std::vector<void*> allocs;
  float *conv1_w = nullptr, *conv1_b = nullptr, *conv2_w = nullptr, *conv2_b = nullptr, *fsq_w = nullptr, *fsq_b = nullptr;
  std::vector<S3Block> blocks;
  float* scratch = nullptr; size_t scratch_floats = 0;
  int32_t* d_ids = nullptr; size_t ids_cap = 0;
};

namespace {

struct S3Loader {
  mia_s3tok* m;
  std::map<std::string, const mia_tensor_view*> by_name;
  std::string err;
  bool get(const std::string& n, std::vector<float>& out, std::initializer_list<int64_t> shp) {
    auto it = by_name.find(n);
    if (it == by_name.end()) { if (err.empty()) err = "missing tensor '" + n + "'"; return false; }
    const mia_tensor_view* t = it->second;
    if (t->dtype != MIA_F32) { if (err.empty()) err = "tensor '" + n + "' must be float32"; return false; }
    bool ok = t->ndim == (int)shp.size(); int i = 0; int64_t numel = 1;
    for (int64_t s : shp) { if (ok && t->shape[i] != s) ok = false; ++i; }
    for (int k = 0; k < t->ndim; ++k) numel *= t->shape[k];
    if (!ok) { if (err.empty()) err = "tensor '" + n + "' has an unexpected shape"; return false; }
    out.assign((const float*)t->data, (const float*)t->data + numel);
    return true;
  }
  float* up(const std::vector<float>& v) {
    void* p = nullptr;
    if (hipMalloc(&p, v.size() * 4 + 64) != hipSuccess) { if (err.empty()) err = "hipMalloc failed"; return nullptr; }
    m->allocs.push_back(p);
    (void)hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    return (float*)p;
  }
  float* vec(const std::string& n, int64_t len) { std::vector<float> v; return get(n, v, {len}) ? up(v) : nullptr; }
  float* mat(const std::string& n, int64_t r, int64_t c) { std::vector<float> v; return get(n, v, {r, c}) ? up(v) : nullptr; }
};

}  // namespace

extern "C" void mia_s3tok_free(mia_s3tok* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) (void)hipFree(p);
  if (m->scratch) (void)hipFree(m->scratch);
  if (m->d_ids) (void)hipFree(m->d_ids);
  delete m;
}

extern "C" mia_s3tok* mia_s3tok_load(mia_ctx* ctx, const mia_s3_config* cfg, const mia_tensor_view* tensors, int n_tensors) {
  if (!ctx) return nullptr;
  auto fail = [&](mia_s3tok* m, const std::string& msg) -> mia_s3tok* { ctx->err = "s3tok_load: " + msg; if (m) mia_s3tok_free(m); return nullptr; };
  if (!cfg || !tensors || n_tensors <= 0) return fail(nullptr, "null arguments");
  const int D = cfg->n_audio_state, H = cfg->n_audio_head, M = cfg->n_mels;
  if (D <= 0 || H * 64 != D || M % 32 || D % 32 || cfg->n_audio_layer <= 0 || D > 4096) return fail(nullptr, "unsupported dims (state = 64*heads, n_mels % 32 == 0)");
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(nullptr, "hipSetDevice failed");
  mia_s3tok* m = new mia_s3tok(); m->ctx = ctx; m->cfg = *cfg;
  S3Loader L; L.m = m;
  for (int i = 0; i < n_tensors; ++i) if (tensors[i].name && tensors[i].data) L.by_name[tensors[i].name] = &tensors[i];
  {  // Conv1d weights [Cout][3][Cin] are already tap-major rows
    std::vector<float> v;
    if (L.get("encoder.conv1.weight", v, {D, 3, M})) m->conv1_w = L.up(v);
    if (L.get("encoder.conv2.weight", v, {D, 3, D})) m->conv2_w = L.up(v);
    m->conv1_b = L.vec("encoder.conv1.bias", D); m->conv2_b = L.vec("encoder.conv2.bias", D);
  }
  m->blocks.resize(cfg->n_audio_layer);
  for (int l = 0; l < cfg->n_audio_layer && L.err.empty(); ++l) {
    const std::string p = "encoder.blocks." + std::to_string(l);
    S3Block& b = m->blocks[l];
    b.attn_ln_g = L.vec(p + ".attn_ln.weight", D); b.attn_ln_b = L.vec(p + ".attn_ln.bias", D);
    b.mlp_ln_g = L.vec(p + ".mlp_ln.weight", D); b.mlp_ln_b = L.vec(p + ".mlp_ln.bias", D);
    std::vector<float> q, k, v, bq, bv;
    if (L.get(p + ".attn.query.weight", q, {D, D}) && L.get(p + ".attn.key.weight", k, {D, D}) && L.get(p + ".attn.value.weight", v, {D, D}) &&
        L.get(p + ".attn.query.bias", bq, {D}) && L.get(p + ".attn.value.bias", bv, {D})) {
      std::vector<float> w((size_t)3 * D * D), bb((size_t)3 * D, 0.f);
      memcpy(w.data(), q.data(), q.size() * 4); memcpy(w.data() + q.size(), k.data(), k.size() * 4); memcpy(w.data() + 2 * q.size(), v.data(), v.size() * 4);
      memcpy(bb.data(), bq.data(), D * 4); memcpy(bb.data() + 2 * D, bv.data(), D * 4);
      b.wqkv = L.up(w); b.bqkv = L.up(bb);
    }
    b.wo = L.mat(p + ".attn.out.weight", D, D); b.bo = L.vec(p + ".attn.out.bias", D);
    {  // depthwise FSMN kernel [D][31][1] -> [31][D]
      std::vector<float> f;
      if (L.get(p + ".attn.fsmn_block.weight", f, {D, 31, 1})) {
        std::vector<float> t((size_t)31 * D);
        for (int c = 0; c < D; ++c) for (int kk = 0; kk < 31; ++kk) t[(size_t)kk * D + c] = f[(size_t)c * 31 + kk];
        b.fsmn = L.up(t);
      }
    }
    b.w1 = L.mat(p + ".mlp.layers.0.weight", 4 * D, D); b.b1 = L.vec(p + ".mlp.layers.0.bias", 4 * D);
    b.w2 = L.mat(p + ".mlp.layers.2.weight", D, 4 * D); b.b2 = L.vec(p + ".mlp.layers.2.bias", D);
  }
  m->fsq_w = L.mat("quantizer.fsq_codebook.project_down.weight", 8, D); m->fsq_b = L.vec("quantizer.fsq_codebook.project_down.bias", 8);
  if (!L.err.empty()) return fail(m, L.err);
  if (hipDeviceSynchronize() != hipSuccess) return fail(m, "device error during upload");
  return m;
}

static int s3_gemm(mia_s3tok* m, const ConvGemmArgs& g) {
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(m->ctx, MIA_ERR_INVALID_ARGUMENT, "%s", e);
  if (codec_conv_gemm_launch(g, 1, m->ctx->stream)) return mia_fail(m->ctx, MIA_ERR_DEVICE, "s3tok: gemm launch failed");
  return MIA_OK;
}

// one clip of T mel frames (T <= 3000: longer audio is windowed by the caller, S3Tokenizer.swift:497-650)
static int s3_encode_clip(mia_s3tok* m, const float* d_mel, int64_t ld_mel, int T, int32_t* d_ids, int* n_tok) {
  mia_ctx* ctx = m->ctx; hipStream_t s = ctx->stream;
  const int D = m->cfg.n_audio_state, H = m->cfg.n_audio_head, NM = m->cfg.n_mels;
  const int T1 = (T + 2 - 2 - 1) / 2 + 1, T2 = (T1 + 2 - 2 - 1) / 2 + 1;     // S3Tokenizer.swift:413-424
  // scratch carve (floats): x0 [(T+2)*NM] | c1 [(T1+2)*D] | x [T2*D] | h [T2*D] | qkv [T2*3D] | att [T2*D] | g [T2*4D]
  const size_t n_x0 = (size_t)(T + 2) * NM, n_c1 = (size_t)(T1 + 2) * D, n_td = (size_t)T2 * D;
  const size_t need = n_x0 + n_c1 + n_td * 2 + n_td * 3 + n_td + n_td * 4 + 1024;
  if (need > m->scratch_floats) {
    MIA_HIP(ctx, hipStreamSynchronize(s));
    if (m->scratch) (void)hipFree(m->scratch);
    m->scratch = nullptr;
    if (hipMalloc((void**)&m->scratch, need * 4) != hipSuccess) return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "s3tok: scratch hipMalloc failed");
    m->scratch_floats = need;
  }
  float* x0 = m->scratch; float* c1 = x0 + n_x0; float* x = c1 + n_c1; float* h = x + n_td; float* qkv = h + n_td; float* att = qkv + 3 * n_td; float* gbuf = att + n_td;
  hipLaunchKernelGGL(s3_transpose_pad, dim3(256), dim3(256), 0, s, d_mel, ld_mel, x0, T, NM);
  MIA_HIP(ctx, hipMemsetAsync(c1, 0, (size_t)D * 4, s));                                  // leading zero row of the conv2 input
  MIA_HIP(ctx, hipMemsetAsync(c1 + (size_t)(T1 + 1) * D, 0, (size_t)D * 4, s));           // trailing zero row
  int rc;
  { ConvGemmArgs g; g.X = x0; g.ldx = NM; g.T_in = T + 2; g.W = m->conv1_w; g.bias = m->conv1_b; g.gelu = 1;
    g.M = T1; g.N = D; g.Cin = NM; g.taps = 3; g.x_row_mul = 2; g.pad = 0; g.Y = c1 + D; g.ldy = D; g.T_out = T1;
    if ((rc = s3_gemm(m, g)) != MIA_OK) return rc; }
  { ConvGemmArgs g; g.X = c1; g.ldx = D; g.T_in = T1 + 2; g.W = m->conv2_w; g.bias = m->conv2_b; g.gelu = 1;
    g.M = T2; g.N = D; g.Cin = D; g.taps = 3; g.x_row_mul = 2; g.pad = 0; g.Y = x; g.ldy = D; g.T_out = T2;
    if ((rc = s3_gemm(m, g)) != MIA_OK) return rc; }
  const float scale = powf(64.0f, -0.25f);
  const size_t att_lds = (size_t)(32 * (((T2 + 63) & ~63) + 1) + 32 * 65 + 64 * 65) * 4;
  if (att_lds > 160 * 1024) return mia_fail(ctx, MIA_ERR_UNSUPPORTED, "s3tok: window too long for the attention kernel (%d tokens)", T2);
  static std::once_flag attr_once;
  std::call_once(attr_once, [] { (void)hipFuncSetAttribute((const void*)s3_attention, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  for (const S3Block& b : m->blocks) {
    if (mia_norm_launch(x, D, b.attn_ln_g, b.attn_ln_b, h, D, T2, D, 1e-5f, false, MIA_F32, s)) return mia_fail(ctx, MIA_ERR_DEVICE, "s3tok: norm launch failed");
    { ConvGemmArgs g; g.X = h; g.ldx = D; g.T_in = T2; g.W = b.wqkv; g.bias = b.bqkv; g.M = T2; g.N = 3 * D; g.Cin = D; g.Y = qkv; g.ldy = 3 * D; g.T_out = T2;
      if ((rc = s3_gemm(m, g)) != MIA_OK) return rc; }
    hipLaunchKernelGGL(s3_fsmn_add, dim3(512), dim3(256), 0, s, x, qkv + 2 * D, (int64_t)3 * D, b.fsmn, T2, D, 31);   // V is still un-touched by RoPE
    hipLaunchKernelGGL(s3_rope_scale, dim3(512), dim3(256), 0, s, qkv, T2, D, 10000.0f, scale);
    hipLaunchKernelGGL(s3_attention, dim3((T2 + 31) / 32, H), dim3(256), att_lds, s, qkv, att, T2, D);
    { ConvGemmArgs g; g.X = att; g.ldx = D; g.T_in = T2; g.W = b.wo; g.bias = b.bo; g.M = T2; g.N = D; g.Cin = D; g.R = x; g.ldr = D; g.Y = x; g.ldy = D; g.T_out = T2;
      if ((rc = s3_gemm(m, g)) != MIA_OK) return rc; }
    if (mia_norm_launch(x, D, b.mlp_ln_g, b.mlp_ln_b, h, D, T2, D, 1e-5f, false, MIA_F32, s)) return mia_fail(ctx, MIA_ERR_DEVICE, "s3tok: norm launch failed");
    { ConvGemmArgs g; g.X = h; g.ldx = D; g.T_in = T2; g.W = b.w1; g.bias = b.b1; g.gelu = 1; g.M = T2; g.N = 4 * D; g.Cin = D; g.Y = gbuf; g.ldy = 4 * D; g.T_out = T2;
      if ((rc = s3_gemm(m, g)) != MIA_OK) return rc; }
    { ConvGemmArgs g; g.X = gbuf; g.ldx = 4 * D; g.T_in = T2; g.W = b.w2; g.bias = b.b2; g.M = T2; g.N = D; g.Cin = 4 * D; g.R = x; g.ldr = D; g.Y = x; g.ldy = D; g.T_out = T2;
      if ((rc = s3_gemm(m, g)) != MIA_OK) return rc; }
  }
  hipLaunchKernelGGL(s3_fsq, dim3((T2 + 3) / 4), dim3(256), 0, s, x, m->fsq_w, m->fsq_b, d_ids, T2, D);
  MIA_HIP(ctx, hipGetLastError());
  *n_tok = T2;
  return MIA_OK;
}

// mel [B][n_mels][T] float32, mel_len [B] (host) -> tokens int32 [B][ceil-ish T/4] zero padded, tok_len [B] (host)
extern "C" int mia_s3tok_encode(mia_s3tok* m, const float* mel, const int32_t* mel_len, int B, int T, int32_t* tokens, int tokens_stride,
                                int32_t* tok_len, int mem) {
  if (!m) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = m->ctx;
  MIA_CHECK_ARG(ctx, mel && mel_len && tokens && tok_len && B > 0 && T > 0, "s3tok_encode: null arguments");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "s3tok_encode: bad mem");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const int NM = m->cfg.n_mels;
  const float* d_mel = mel;
  if (mem == MIA_MEM_HOST) {
    const size_t bytes = (size_t)B * NM * T * 4;
    void* ws = mia_workspace(ctx, bytes);
    if (!ws) return MIA_ERR_OUT_OF_MEMORY;
    MIA_HIP(ctx, hipMemcpyAsync(ws, mel, bytes, hipMemcpyHostToDevice, ctx->stream));
    d_mel = (const float*)ws;
  }
  if ((size_t)tokens_stride > m->ids_cap) {
    MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (m->d_ids) (void)hipFree(m->d_ids);
    m->d_ids = nullptr;
    if (hipMalloc((void**)&m->d_ids, (size_t)tokens_stride * 4 + 64) != hipSuccess) return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "s3tok: hipMalloc failed");
    m->ids_cap = tokens_stride;
  }
  for (int b = 0; b < B; ++b) {
    const int L = mel_len[b];
    MIA_CHECK_ARG(ctx, L > 0 && L <= T, "s3tok_encode: mel_len[%d] = %d out of range", b, L);
    MIA_CHECK_ARG(ctx, L <= 3000, "s3tok_encode: clip %d has %d frames; windows above 30 s are split by the caller (S3Tokenizer.swift:497-650)", b, L);
    int n = 0;
    int32_t* dst = mem == MIA_MEM_DEVICE ? tokens + (size_t)b * tokens_stride : m->d_ids;
    MIA_CHECK_ARG(ctx, ((L - 1) / 2 + 1 - 1) / 2 + 1 <= tokens_stride, "s3tok_encode: tokens_stride too small");
    MIA_HIP(ctx, hipMemsetAsync(dst, 0, (size_t)tokens_stride * 4, ctx->stream));
    int rc = s3_encode_clip(m, d_mel + (size_t)b * NM * T, T, L, dst, &n);
    if (rc != MIA_OK) return rc;
    tok_len[b] = n;
    if (mem == MIA_MEM_HOST) {
      MIA_HIP(ctx, hipMemcpyAsync(tokens + (size_t)b * tokens_stride, m->d_ids, (size_t)tokens_stride * 4, hipMemcpyDeviceToHost, ctx->stream));
      MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  return MIA_OK;
}
