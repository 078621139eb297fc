// flow.hip -- CosyVoice2 token -> mel flow on gfx950, fp32.
//
// Replaces CosyVoice2FlowModule.inference (TTS/CosyVoice2/CosyVoice2Model.swift:467-553): input_embedding ->
// UpsampleConformerEncoder (Codec/S3Gen/Transformer/UpsampleConformerEncoder.swift:407-474) -> encoder_proj ->
// CosyVoice2ConditionalCFM (TTS/CosyVoice2/Flow/CosyVoice2CFM.swift:74-187: cosine schedule, Euler, classifier-free guidance with
// the cond / uncond pair stacked on the batch axis) around the ConditionalDecoder estimator (Codec/S3Gen/S3GenDecoder.swift:277-400).
//
// Layout: activations are time-major fp32 [T][C]; the CFG pair is two stacked sequences [2 T][C] (tap GEMMs get seg = T so causal
// convolutions never read across the pair).  The estimator input [x | mu | spks | cond] is one persistent [2 T][320] buffer: only
// its first 80 columns change per Euler step and the Euler update writes them in place.  Every Linear / Conv1d runs on the
// exact-fp32 MFMA tap GEMM (codec_kernels.hip) with bias, SiLU / GELU / leaky-ReLU and residual adds in its epilogue; attention
// is attn_f32.hip.  The reference engine drives this path with batch 1 and full-length masks (all ones): that is what is built.
// The CFM's initial noise z is an explicit input (the reference draws it with MLXRandom.normal, CosyVoice2CFM.swift:86).
#include <cmath>
#include <string>
#include <vector>

#include "codec.h"
#include "mia_device.h"
#include "mia_internal.h"
#include "ops.h"
#include "tensor_loader.h"

namespace {

struct Lin { float* w = nullptr; float* b = nullptr; int N = 0, K = 0, taps = 1; };
struct Norm { float* g = nullptr; float* b = nullptr; };
struct ConfLayer { Norm n_mha, n_ff; Lin qkv, pos, out, ff1, ff2; float* u = nullptr; float* v = nullptr; };
struct TBlock { Norm n1, n3; Lin qkv, out, ff1, ff2; };
struct Resnet { Lin c1, c2, res; Norm n1, n2; int idx = 0; };
struct UBlock { Resnet rn; std::vector<TBlock> tb; };

// rows of the embedding table, ids clipped to the table (CosyVoice2Model.swift:496-501)
__global__ __launch_bounds__(256) void flow_gather(const int32_t* __restrict__ ids, const float* __restrict__ table, float* __restrict__ out,
                                                   int n, int D, int V) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d4 = D >> 2;
  if (e >= (int64_t)n * d4) return;
  const int r = (int)(e / d4), c = (int)(e % d4) * 4;
  int id = ids[r]; id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  *reinterpret_cast<float4*>(out + (int64_t)r * D + c) = *reinterpret_cast<const float4*>(table + (int64_t)id * D + c);
}

// nearest-neighbour upsampling: out[t] = x[t / stride]   (Upsample1D, UpsampleConformerEncoder.swift:41)
__global__ __launch_bounds__(256) void flow_repeat_rows(const float* __restrict__ x, float* __restrict__ out, int T_out, int D, int stride) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int d4 = D >> 2;
  if (e >= (int64_t)T_out * d4) return;
  const int t = (int)(e / d4), c = (int)(e % d4) * 4;
  *reinterpret_cast<float4*>(out + (int64_t)t * D + c) = *reinterpret_cast<const float4*>(x + (int64_t)(t / stride) * D + c);
}

// PositionalEncoding.createPE (Embedding.swift:33-52): pe[t][2 i] = sin(t w_i), pe[t][2 i + 1] = cos(t w_i), w_i = exp(2 i * (-ln 1e4 / D))
__global__ __launch_bounds__(256) void flow_sinusoid(float* __restrict__ pe, int T, int D, float neg_log_over_d) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int half = D >> 1;
  if (e >= (int64_t)T * half) return;
  const int t = (int)(e / half), i = (int)(e % half);
  const float div = expf((float)(2 * i) * neg_log_over_d);
  const float arg = (float)t * div;
  pe[(int64_t)t * D + 2 * i] = sinf(arg);
  pe[(int64_t)t * D + 2 * i + 1] = cosf(arg);
}

// spks = Linear(e / (||e|| + 1e-8))   (CosyVoice2Model.swift:479-482); one block
__global__ __launch_bounds__(256) void flow_spks(const float* __restrict__ emb, const float* __restrict__ w, const float* __restrict__ b,
                                                 float* __restrict__ out, int E, int M) {
  __shared__ float red[4];
  __shared__ float en[1024];
  float s = 0.f;
  for (int i = threadIdx.x; i < E; i += 256) { const float x = emb[i]; s += x * x; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
  for (int i = threadIdx.x; i < E; i += 256) en[i] = emb[i] / (nrm + 1e-8f);
  __syncthreads();
  for (int m = threadIdx.x; m < M; m += 256) {
    float acc = b[m];
    for (int i = 0; i < E; ++i) acc = fmaf(w[(int64_t)m * E + i], en[i], acc);
    out[m] = acc;
  }
}

// static 240 columns of the estimator input of ONE utterance: hin[0][t] = [ . | mu[t] | spks | cond[t] ], hin[1][t] = [ . | 0 | 0 | 0 ];
// cond[t] = prompt_feat[t] for t < m1, else 0; and x = z^T into columns 0..79 of both halves.  The utterance owns `rows` >= T rows per
// half (stacked, padded batch): rows T .. rows-1 are written as zeros, so every row of the stack is defined.
__global__ __launch_bounds__(256) void flow_pack_inputs(const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ spks,
                                                        const float* __restrict__ pf, float* __restrict__ hin, float* __restrict__ x, int T, int M,
                                                        int m1, int rows) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)rows * M) return;
  const int t = (int)(e / M), c = (int)(e % M);
  const int64_t ld = 4 * M;
  float* r0 = hin + (int64_t)t * ld;
  float* r1 = hin + ((int64_t)rows + t) * ld;
  const bool in = t < T;
  const float xv = in ? z[(int64_t)c * T + t] : 0.f;
  x[(int64_t)t * M + c] = xv;
  r0[c] = xv; r1[c] = xv;
  r0[M + c] = in ? mu[(int64_t)t * M + c] : 0.f; r1[M + c] = 0.f;
  r0[2 * M + c] = in ? spks[c] : 0.f; r1[2 * M + c] = 0.f;
  r0[3 * M + c] = t < m1 ? pf[(int64_t)t * M + c] : 0.f; r1[3 * M + c] = 0.f;
}

// x += dt ((1 + r) d_cond - r d_uncond), mirrored into columns 0..79 of both halves of hin   (CosyVoice2CFM.swift:166-176);
// blockIdx.y = utterance: d / hin hold [cond | uncond] halves of T rows each per utterance, x holds T rows per utterance
__global__ __launch_bounds__(256) void flow_euler(const float* __restrict__ d, float* __restrict__ x, float* __restrict__ hin, int T, int M,
                                                  float dt, float rate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)T * M) return;
  const int t = (int)(e / M), c = (int)(e % M);
  const int64_t u = blockIdx.y;
  d += u * 2 * T * M; hin += u * 2 * T * 4 * M; x += u * T * M;
  const float dc = d[(int64_t)t * M + c], du = d[((int64_t)T + t) * M + c];
  const float comb = (1.0f + rate) * dc - rate * du;
  const float xv = x[e] + dt * comb;
  x[e] = xv;
  hin[(int64_t)t * 4 * M + c] = xv;
  hin[((int64_t)T + t) * 4 * M + c] = xv;
}

// out[c][t - m1] = x[t][c]
__global__ __launch_bounds__(256) void flow_emit(const float* __restrict__ x, float* __restrict__ out, int T, int M, int m1) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int To = T - m1;
  if (e >= (int64_t)To * M) return;
  const int c = (int)(e / To), t = (int)(e % To);
  out[e] = x[(int64_t)(t + m1) * M + c];
}

__device__ __forceinline__ float mish_f(float x) { return x * tanhf(logf(1.0f + expf(x))); }

__global__ __launch_bounds__(256) void flow_mish(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) y[e] = mish_f(x[e]);
}

// y = mish(LayerNorm(x)) (+ add[c])   (CausalBlock1D, S3GenDecoder.swift:62-70; the time-embedding add of CausalResnetBlock1D :91);
// one wave per row, C <= 512, C % 4 == 0
__global__ __launch_bounds__(256) void flow_ln_mish(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                    const float* __restrict__ add, float* __restrict__ y, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * C;
  const int nv = C >> 2;
  f32x4 v[2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) { v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (lane + 64 * i < nv)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float dd = v[i][j] - mean; q += dd * dd; }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + 64 * i;
    if (c >= nv) continue;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(g + 4 * c);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(b + 4 * c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = mish_f((v[i][j] - mean) * rstd * gm[j] + bt[j]);
      if (add) o[j] += add[4 * c + j];
    }
    *reinterpret_cast<f32x4*>(y + (int64_t)row * C + 4 * c) = o;
  }
}

}  // namespace

struct mia_flow {
  mia_ctx* ctx = nullptr;
  mia_flow_config cfg{};
  std::vector<void*> allocs;
  float* emb_table = nullptr;
  Lin spk, embed, up_embed, pl1, pl2, up_conv, enc_proj;
  Norm embed_n, up_embed_n, after_n;       // embed norms carry the sqrt(D) xscale folded in
  std::vector<ConfLayer> enc, up_enc;
  Lin t1, t2, tproj;                        // time MLP; tproj = every resnet's mlp_linear stacked [n_res * C][TE]
  float* tproj_b = nullptr;
  UBlock down, up;
  std::vector<UBlock> mid;
  Lin down_conv, up_conv2, final_conv, final_proj;
  Norm final_n;
  int n_res = 0;
  float* arena = nullptr; size_t arena_floats = 0;
  int32_t* d_ids = nullptr; size_t ids_cap = 0;
};

namespace {

struct FLoader : TensorLoader {
  bool lin(const std::string& p, int N, int K, bool bias, Lin& o) {
    std::vector<float> w, b;
    if (!f32(p + ".weight", w, {N, K})) return false;
    o.w = up(w); o.N = N; o.K = K; o.taps = 1;
    if (bias) { if (!f32(p + ".bias", b, {N})) return false; o.b = up(b); }
    return true;
  }
  bool conv(const std::string& p, int N, int taps, int K, Lin& o) {
    std::vector<float> w, b;
    if (!f32(p + ".weight", w, {N, taps, K}) || !f32(p + ".bias", b, {N})) return false;
    o.w = up(w); o.b = up(b); o.N = N; o.K = K; o.taps = taps;
    return true;
  }
  bool norm(const std::string& p, int D, Norm& o, float scale = 1.0f) {
    std::vector<float> g, b;
    if (!f32(p + ".weight", g, {D}) || !f32(p + ".bias", b, {D})) return false;
    if (scale != 1.0f) { for (float& x : g) x *= scale; for (float& x : b) x *= scale; }
    o.g = up(g); o.b = up(b);
    return true;
  }
  // several [N_i][K] Linear layers stacked along N (shared input): fused q|k|v projections
  bool stacked(const std::vector<std::string>& ps, int N, int K, bool bias, const std::vector<bool>& has_bias, Lin& o) {
    std::vector<float> w, b;
    for (size_t i = 0; i < ps.size(); ++i) {
      std::vector<float> wi, bi;
      if (!f32(ps[i] + ".weight", wi, {N, K})) return false;
      w.insert(w.end(), wi.begin(), wi.end());
      if (bias) {
        if (has_bias[i] && has(ps[i] + ".bias")) { if (!f32(ps[i] + ".bias", bi, {N})) return false; }
        else bi.assign(N, 0.f);
        b.insert(b.end(), bi.begin(), bi.end());
      }
    }
    o.w = up(w); o.N = N * (int)ps.size(); o.K = K; o.taps = 1;
    if (bias) o.b = up(b);
    return true;
  }
  bool conformer(const std::string& p, int D, int H, int FF, ConfLayer& l) {
    const std::string a = p + ".self_attn";
    if (!norm(p + ".norm_mha", D, l.n_mha) || !norm(p + ".norm_ff", D, l.n_ff)) return false;
    if (!stacked({a + ".linear_q", a + ".linear_k", a + ".linear_v"}, D, D, true, {true, true, true}, l.qkv)) return false;
    if (!lin(a + ".linear_pos", D, D, false, l.pos) || !lin(a + ".linear_out", D, D, true, l.out)) return false;
    std::vector<float> u, v;
    if (!f32(a + ".pos_bias_u", u, {H, D / H}) || !f32(a + ".pos_bias_v", v, {H, D / H})) return false;
    l.u = up(u); l.v = up(v);
    return lin(p + ".feed_forward.w_1", FF, D, true, l.ff1) && lin(p + ".feed_forward.w_2", D, FF, true, l.ff2);
  }
  bool tblock(const std::string& p, int C, int inner, TBlock& t) {
    const std::string a = p + ".attn";
    if (!norm(p + ".norm1", C, t.n1) || !norm(p + ".norm3", C, t.n3)) return false;
    if (!stacked({a + ".query_proj", a + ".key_proj", a + ".value_proj"}, inner, C, false, {false, false, false}, t.qkv)) return false;
    return lin(a + ".out_proj", C, inner, true, t.out) && lin(p + ".ff.layers.0", 4 * C, C, true, t.ff1) &&
           lin(p + ".ff.layers.1", C, 4 * C, true, t.ff2);
  }
  bool resnet(const std::string& p, int Cin, int C, int TE, Resnet& r, std::vector<float>& tw, std::vector<float>& tb, int idx) {
    std::vector<float> w, b;
    if (!f32(p + ".mlp_linear.weight", w, {C, TE}) || !f32(p + ".mlp_linear.bias", b, {C})) return false;
    tw.insert(tw.end(), w.begin(), w.end()); tb.insert(tb.end(), b.begin(), b.end());
    r.idx = idx;
    return conv(p + ".block1.conv.conv", C, 3, Cin, r.c1) && norm(p + ".block1.norm", C, r.n1) && conv(p + ".block2.conv.conv", C, 3, C, r.c2) &&
           norm(p + ".block2.norm", C, r.n2) && conv(p + ".res_conv", C, 1, Cin, r.res);
  }
  bool ublock(const std::string& p, int Cin, int C, int inner, int TE, int nb, UBlock& u, std::vector<float>& tw, std::vector<float>& tb, int idx) {
    if (!resnet(p + ".resnet", Cin, C, TE, u.rn, tw, tb, idx)) return false;
    u.tb.resize(nb);
    for (int j = 0; j < nb; ++j) if (!tblock(p + ".transformers." + std::to_string(j), C, inner, u.tb[j])) return false;
    return true;
  }
};

struct Run {
  mia_flow* f; hipStream_t s; int rc = MIA_OK;
  bool fail(int code, const char* msg) { if (rc == MIA_OK) rc = mia_fail(f->ctx, code, "flow: %s", msg); return false; }

  // Y[M][N] = act(X[M][K*taps] W^T + b) (+ R); taps > 1: window starting `pad` rows back (zero outside the sequence); nseq > 1: X / Y / R
  // are nseq stacked sequences of M rows each, convolved independently (one grid.z phase per sequence)
  bool gemm(const Lin& l, const float* X, int64_t ldx, int M, float* Y, int64_t ldy, int act = 0, const float* R = nullptr, int pad = 0,
            int nseq = 1, const int32_t* phase_len = nullptr) {
    if (rc != MIA_OK) return false;
    ConvGemmArgs g;
    g.X = X; g.ldx = ldx; g.T_in = M; g.W = l.w; g.bias = l.b; g.Y = Y; g.ldy = ldy; g.T_out = M * nseq; g.R = R; g.ldr = ldy;
    g.M = M; g.N = l.N; g.Cin = l.K; g.taps = l.taps; g.pad = pad; g.gelu = act;
    if (nseq > 1) { g.x_phase_step = M; g.y_phase_step = M; g.phase_len = phase_len; }
    if (const char* e = codec_conv_gemm_check(g)) return fail(MIA_ERR_INVALID_ARGUMENT, e);
    if (codec_conv_gemm_launch(g, nseq, s)) return fail(MIA_ERR_DEVICE, "gemm launch failed");
    return true;
  }
  bool ln(const Norm& n, const float* x, float* y, int M, int D, float eps) {
    if (rc != MIA_OK) return false;
    if (mia_norm_launch(x, D, n.g, n.b, y, D, M, D, eps, false, MIA_F32, s)) return fail(MIA_ERR_DEVICE, "norm launch failed");
    return true;
  }
  bool ln_mish(const Norm& n, const float* x, const float* add, float* y, int M, int C) {
    if (rc != MIA_OK) return false;
    hipLaunchKernelGGL(flow_ln_mish, dim3((M + 3) / 4), dim3(256), 0, s, x, n.g, n.b, add, y, M, C, 1e-5f);
    return true;
  }
  bool attn(const AttnF32Args& a) {
    if (rc != MIA_OK) return false;
    if (const char* e = mia_attn_f32_check(a)) return fail(MIA_ERR_INVALID_ARGUMENT, e);
    if (mia_attn_f32_launch(a, s)) return fail(MIA_ERR_DEVICE, "attention launch failed");
    return true;
  }
};

struct Bufs {
  float *x0, *x, *h, *qkv, *pb, *att, *g, *pe, *mu, *spks, *z, *pf, *emb;       // encoder side
  float *hin, *xs, *c1, *h1, *xr, *att2, *d, *tsin, *te1, *te2, *tm, *tpr, *out, *xr2, *cat;  // estimator side
};

// T = rows of the (stacked) encoder output: U utterances x the longest one's frames
size_t carve(mia_flow* f, int Tt, int T, int S, Bufs* b, int U = 1) {
  const mia_flow_config& c = f->cfg;
  const size_t D = c.input_size, FF = c.enc_linear_units, M = c.output_size, C = c.dec_channels, inner = (size_t)c.dec_heads * 64;
  const size_t TE = 4 * C;
  auto al = [](size_t n) { return (n + 63) / 64 * 64; };
  const size_t T2 = 2 * (size_t)T;
  size_t sizes[] = {
      al((size_t)T * D), al((size_t)T * D), al((size_t)T * D), al((size_t)T * 3 * D), al((size_t)T * D), al((size_t)T * D), al((size_t)T * FF),
      al((size_t)T * D), al((size_t)T * M), al((size_t)U * M), al((size_t)T * M), al((size_t)T * M), al((size_t)U * c.spk_embed_dim),
      al(T2 * 4 * M), al((size_t)T * M), al(T2 * C), al(T2 * std::max(std::max(3 * inner, 4 * C), 2 * C)), al(T2 * C), al(T2 * inner), al(T2 * M),
      al((size_t)S * c.dec_in_channels), al((size_t)S * TE), al((size_t)S * TE), al((size_t)S * TE), al((size_t)S * f->n_res * C), al((size_t)T * M),
      al(T2 * C), al(T2 * 2 * C)};
  size_t tot = 0;
  for (size_t s : sizes) tot += s;
  if (!b) return tot;
  float* q = f->arena;
  float** dst[] = {&b->x0, &b->x, &b->h, &b->qkv, &b->pb, &b->att, &b->g, &b->pe, &b->mu, &b->spks, &b->z, &b->pf, &b->emb,
                   &b->hin, &b->xs, &b->c1, &b->h1, &b->xr, &b->att2, &b->d, &b->tsin, &b->te1, &b->te2, &b->tm, &b->tpr, &b->out, &b->xr2, &b->cat};
  for (int i = 0; i < 28; ++i) { *dst[i] = q; q += sizes[i]; }
  (void)Tt;
  return tot;
}

// one ConformerEncoderLayer (ConformerEncoderLayer.swift:69-165): pre-norm rel-pos MHA + pre-norm SiLU FFN, on U stacked sequences of
// T rows each (len: their valid rows on the device, null when U == 1)
void conformer_layer(Run& r, const ConfLayer& l, Bufs& b, int T, int D, int H, int FF, int chunk = 0, int U = 1, const int32_t* len = nullptr) {
  const int R = U * T;
  r.ln(l.n_mha, b.x, b.h, R, D, 1e-12f);
  r.gemm(l.qkv, b.h, D, R, b.qkv, 3 * D);
  r.gemm(l.pos, b.pe, D, T, b.pb, D);
  AttnF32Args a;
  a.q = b.qkv; a.ldq = 3 * D; a.k = b.qkv + D; a.ldk = 3 * D; a.v = b.qkv + 2 * D; a.ldv = 3 * D; a.p = b.pb; a.ldp = D;
  a.bias_u = l.u; a.bias_v = l.v; a.out = b.att; a.ldo = D; a.B = U; a.T = T; a.H = H; a.scale = 1.0f / sqrtf((float)(D / H));
  a.chunk = chunk; a.seq_len = len;
  r.attn(a);
  r.gemm(l.out, b.att, D, R, b.x, D, 0, b.x);
  r.ln(l.n_ff, b.x, b.h, R, D, 1e-12f);
  r.gemm(l.ff1, b.h, D, R, b.g, FF, 4);
  r.gemm(l.ff2, b.g, FF, R, b.x, D, 0, b.x);
}

// tokens (device, U stacked sequences of Tt ids each, padded) -> mu [U][T][80]; len_tok / len_up: valid tokens / frames per sequence
int run_encoder(mia_flow* f, Bufs& b, const int32_t* d_ids, int Tt, int chunk = 0, int U = 1, const int32_t* len_tok = nullptr,
                const int32_t* len_up = nullptr) {
  const mia_flow_config& c = f->cfg;
  const int D = c.input_size, H = c.enc_heads, FF = c.enc_linear_units, st = c.upsample_stride, T = Tt * st;
  const int Rt = U * Tt, R = U * T;
  Run r{f, f->ctx->stream};
  hipStream_t s = r.s;
  hipLaunchKernelGGL(flow_gather, dim3((unsigned)(((int64_t)Rt * (D / 4) + 255) / 256)), dim3(256), 0, s, d_ids, f->emb_table, b.x0, Rt, D, c.vocab_size);
  hipLaunchKernelGGL(flow_sinusoid, dim3((unsigned)(((int64_t)T * (D / 2) + 255) / 256)), dim3(256), 0, s, b.pe, T, D, -logf(10000.0f) / (float)D);
  // LinearNoSubsampling: linear -> LayerNorm(1e-5) -> * sqrt(D) (folded into the norm's affine)
  r.gemm(f->embed, b.x0, D, Rt, b.h, D);
  r.ln(f->embed_n, b.h, b.x0, Rt, D, 1e-5f);
  // PreLookaheadLayer: conv1 over x[t .. t+L] (zero beyond the sequence's end) -> leaky-ReLU -> causal conv2 (k 3) -> + x
  r.gemm(f->pl1, b.x0, D, Tt, b.h, D, 5, nullptr, 0, U, len_tok);
  r.gemm(f->pl2, b.h, D, Tt, b.x, D, 0, b.x0, 2, U);
  for (const ConfLayer& l : f->enc) conformer_layer(r, l, b, Tt, D, H, FF, chunk, U, len_tok);
  // Upsample1D: nearest repeat x stride, left pad 2 stride, conv k = 2 stride + 1
  hipLaunchKernelGGL(flow_repeat_rows, dim3((unsigned)(((int64_t)R * (D / 4) + 255) / 256)), dim3(256), 0, s, b.x, b.att, R, D, st);
  r.gemm(f->up_conv, b.att, D, T, b.h, D, 0, nullptr, 2 * st, U);
  r.gemm(f->up_embed, b.h, D, R, b.x0, D);
  r.ln(f->up_embed_n, b.x0, b.x, R, D, 1e-5f);
  for (const ConfLayer& l : f->up_enc) conformer_layer(r, l, b, T, D, H, FF, chunk * st, U, len_up);      // effectiveUpChunkSize (:452)
  r.ln(f->after_n, b.x, b.h, R, D, 1e-5f);
  r.gemm(f->enc_proj, b.h, D, R, b.mu, c.output_size);
  if (r.rc == MIA_OK && hipGetLastError() != hipSuccess) return mia_fail(f->ctx, MIA_ERR_DEVICE, "flow: encoder launch failed");
  return r.rc;
}

// CausalResnetBlock1D (S3GenDecoder.swift:89-101) on ns stacked sequences of T rows: X [ns T][Cin] -> Y [ns T][C]  (Y must not alias X)
void resnet(Run& r, const Resnet& rn, Bufs& b, const float* X, int Cin, int T, int C, const float* tvec, float* Y, int ns = 2) {
  const int M = ns * T;
  r.gemm(rn.c1, X, Cin, T, b.c1, C, 0, nullptr, 2, ns);
  r.ln_mish(rn.n1, b.c1, tvec + (size_t)rn.idx * C, b.h1, M, C);
  r.gemm(rn.c2, b.h1, C, T, b.c1, C, 0, nullptr, 2, ns);
  r.ln_mish(rn.n2, b.c1, nullptr, b.h1, M, C);
  r.gemm(rn.res, X, Cin, M, Y, C, 0, b.h1);
}

// BasicTransformerBlock (MatchaTransformer.swift:128-146) in place on x [ns T][C]; len: valid rows per sequence (device) or null
void tblock(Run& r, const TBlock& t, Bufs& b, float* x, int T, int C, int H, int chunk = 0, int ns = 2, const int32_t* len = nullptr) {
  const int M = ns * T, inner = H * 64;
  float* h = b.c1;          // [M][C] scratch
  float* big = b.h1;        // [M][max(3 inner, 4 C)] scratch
  r.ln(t.n1, x, h, M, C, 1e-5f);
  r.gemm(t.qkv, h, C, M, big, 3 * inner);
  AttnF32Args a;
  a.q = big; a.ldq = 3 * inner; a.k = big + inner; a.ldk = 3 * inner; a.v = big + 2 * inner; a.ldv = 3 * inner;
  a.out = b.att2; a.ldo = inner; a.B = ns; a.T = T; a.H = H; a.scale = 0.125f; a.chunk = chunk; a.seq_len = len;
  r.attn(a);
  r.gemm(t.out, b.att2, inner, M, x, C, 0, x);
  r.ln(t.n3, x, h, M, C, 1e-5f);
  r.gemm(t.ff1, h, C, M, big, 4 * C, 1);
  r.gemm(t.ff2, big, 4 * C, M, x, C, 0, x);
}

}  // namespace

extern "C" {

mia_flow* mia_flow_load(mia_ctx* ctx, const mia_flow_config* cfg, const mia_tensor_view* tensors, int n_tensors) {
  if (!ctx) return nullptr;
  if (!cfg || !tensors) { mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "flow_load: null argument"); return nullptr; }
  const mia_flow_config& c = *cfg;
  const bool ok_cfg = c.input_size % 32 == 0 && c.input_size <= 512 && c.enc_heads > 0 && c.input_size == c.enc_heads * 64 && c.output_size == 80 &&
                      c.dec_in_channels == 4 * c.output_size && c.dec_channels % 32 == 0 && c.dec_channels <= 512 && c.dec_heads > 0 &&
                      c.enc_linear_units % 32 == 0 && c.spk_embed_dim > 0 && c.spk_embed_dim <= 1024 && c.upsample_stride >= 1 &&
                      c.pre_lookahead_len >= 0 && c.enc_blocks >= 0 && c.enc_up_blocks >= 0 && c.dec_n_blocks >= 0 && c.dec_mid_blocks >= 0 &&
                      c.vocab_size > 0 && c.n_timesteps > 0;
  if (!ok_cfg) { mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "flow_load: unsupported configuration (head dim must be 64, mel 80)"); return nullptr; }
  if (hipSetDevice(ctx->device) != hipSuccess) { mia_fail(ctx, MIA_ERR_DEVICE, "flow_load: hipSetDevice failed"); return nullptr; }
  mia_flow* f = new mia_flow();
  f->ctx = ctx; f->cfg = c;
  FLoader L; L.allocs = &f->allocs; L.index(tensors, n_tensors);
  const int D = c.input_size, M = c.output_size, C = c.dec_channels, TE = 4 * C, inner = c.dec_heads * 64;
  bool ok = true;
  { std::vector<float> t; ok = L.f32("input_embedding.weight", t, {c.vocab_size, D}); if (ok) f->emb_table = L.up(t); }
  ok = ok && L.lin("spk_embed_affine_layer", M, c.spk_embed_dim, true, f->spk);
  const float xs = sqrtf((float)D);
  ok = ok && L.lin("encoder.embed.linear", D, D, true, f->embed) && L.norm("encoder.embed.norm", D, f->embed_n, xs);
  ok = ok && L.lin("encoder.up_embed.linear", D, D, true, f->up_embed) && L.norm("encoder.up_embed.norm", D, f->up_embed_n, xs);
  ok = ok && L.conv("encoder.pre_lookahead_layer.conv1", D, c.pre_lookahead_len + 1, D, f->pl1) && L.conv("encoder.pre_lookahead_layer.conv2", D, 3, D, f->pl2);
  ok = ok && L.conv("encoder.up_layer.conv", D, 2 * c.upsample_stride + 1, D, f->up_conv);
  f->enc.resize(c.enc_blocks); f->up_enc.resize(c.enc_up_blocks);
  for (int i = 0; i < c.enc_blocks && ok; ++i) ok = L.conformer("encoder.encoders." + std::to_string(i), D, c.enc_heads, c.enc_linear_units, f->enc[i]);
  for (int i = 0; i < c.enc_up_blocks && ok; ++i) ok = L.conformer("encoder.up_encoders." + std::to_string(i), D, c.enc_heads, c.enc_linear_units, f->up_enc[i]);
  ok = ok && L.norm("encoder.after_norm", D, f->after_n) && L.lin("encoder_proj", M, D, true, f->enc_proj);
  const std::string e = "decoder.estimator";
  ok = ok && L.lin(e + ".time_mlp.linear_1", TE, c.dec_in_channels, true, f->t1) && L.lin(e + ".time_mlp.linear_2", TE, TE, true, f->t2);
  std::vector<float> tw, tb;
  int idx = 0;
  ok = ok && L.ublock(e + ".down_blocks.0", c.dec_in_channels, C, inner, TE, c.dec_n_blocks, f->down, tw, tb, idx++);
  ok = ok && L.conv(e + ".down_blocks.0.downsample.conv", C, 3, C, f->down_conv);
  f->mid.resize(c.dec_mid_blocks);
  for (int i = 0; i < c.dec_mid_blocks && ok; ++i) ok = L.ublock(e + ".mid_blocks." + std::to_string(i), C, C, inner, TE, c.dec_n_blocks, f->mid[i], tw, tb, idx++);
  ok = ok && L.ublock(e + ".up_blocks.0", 2 * C, C, inner, TE, c.dec_n_blocks, f->up, tw, tb, idx++);
  ok = ok && L.conv(e + ".up_blocks.0.upsample.conv", C, 3, C, f->up_conv2);
  ok = ok && L.conv(e + ".final_block.conv.conv", C, 3, C, f->final_conv) && L.norm(e + ".final_block.norm", C, f->final_n);
  ok = ok && L.conv(e + ".final_proj", M, 1, C, f->final_proj);
  if (ok) { f->n_res = idx; f->tproj.w = L.up(tw); f->tproj.b = L.up(tb); f->tproj.N = idx * C; f->tproj.K = TE; }
  if (!ok || !L.err.empty()) {
    mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "flow_load: %s", L.err.empty() ? "failed" : L.err.c_str());
    mia_flow_free(f);
    return nullptr;
  }
  return f;
}

void mia_flow_free(mia_flow* f) {
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  for (void* p : f->allocs) (void)hipFree(p);
  if (f->arena) (void)hipFree(f->arena);
  if (f->d_ids) (void)hipFree(f->d_ids);
  delete f;
}

// scratch for U stacked utterances of at most Tt tokens; the id buffer also holds the 4 U per-sequence lengths behind the ids
static int flow_prepare(mia_flow* f, int Tt, int S, Bufs& b, int U = 1) {
  const int T = Tt * f->cfg.upsample_stride;
  const size_t need = carve(f, Tt, U * T, S, nullptr, U);
  if (need > f->arena_floats) {
    MIA_HIP(f->ctx, hipStreamSynchronize(f->ctx->stream));
    if (f->arena) (void)hipFree(f->arena);
    f->arena = nullptr; f->arena_floats = 0;
    if (hipMalloc((void**)&f->arena, need * 4) != hipSuccess) return mia_fail(f->ctx, MIA_ERR_OUT_OF_MEMORY, "flow: scratch hipMalloc failed");
    f->arena_floats = need;
  }
  carve(f, Tt, U * T, S, &b, U);
  const size_t ids = (size_t)U * Tt + 4 * (size_t)U;
  if (ids > f->ids_cap) {
    MIA_HIP(f->ctx, hipStreamSynchronize(f->ctx->stream));
    if (f->d_ids) (void)hipFree(f->d_ids);
    f->d_ids = nullptr; f->ids_cap = 0;
    if (hipMalloc((void**)&f->d_ids, ids * 4 + 64) != hipSuccess) return mia_fail(f->ctx, MIA_ERR_OUT_OF_MEMORY, "flow: hipMalloc failed");
    f->ids_cap = ids;
  }
  return MIA_OK;
}

static int upload_tokens(mia_flow* f, int32_t* dst, const int32_t* prompt_token, int n_prompt, const int32_t* token, int n_token, int mem) {
  const hipMemcpyKind kind = mem == MIA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (n_prompt) MIA_HIP(f->ctx, hipMemcpyAsync(dst, prompt_token, (size_t)n_prompt * 4, kind, f->ctx->stream));
  MIA_HIP(f->ctx, hipMemcpyAsync(dst + n_prompt, token, (size_t)n_token * 4, kind, f->ctx->stream));
  return MIA_OK;
}

int mia_flow_encode(mia_flow* f, const int32_t* token, int n_token, float* mu, int mem) {
  if (!f) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_CHECK_ARG(f->ctx, token && mu && n_token > 0 && n_token <= 4096 && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "flow_encode: bad argument");
  MIA_HIP(f->ctx, hipSetDevice(f->ctx->device));
  Bufs b;
  if (int rc = flow_prepare(f, n_token, 1, b)) return rc;
  if (int rc = upload_tokens(f, f->d_ids, nullptr, 0, token, n_token, mem)) return rc;
  if (int rc = run_encoder(f, b, f->d_ids, n_token)) return rc;
  const size_t n = (size_t)n_token * f->cfg.upsample_stride * f->cfg.output_size;
  MIA_HIP(f->ctx, hipMemcpyAsync(mu, b.mu, n * 4, mem == MIA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, f->ctx->stream));
  if (mem == MIA_MEM_HOST) MIA_HIP(f->ctx, hipStreamSynchronize(f->ctx->stream));
  return MIA_OK;
}

// One utterance of a flow call (all pointers host or device per `mem`)
struct FlowUtt {
  const int32_t* token; int n_token; const int32_t* prompt_token; int n_prompt; const float* prompt_feat; int prompt_feat_len;
  const float* embedding; const float* z; float* mel; int* mel_frames;
};

// U utterances side by side: every stage runs once on the stack of U (x 2 for the estimator's cond / uncond pair) sequences, each padded
// to the longest one's length.  Padding never reaches a valid row: the convolutions are causal except the encoder's look-ahead window
// (per-sequence zero beyond the end, ConvGemmArgs::phase_len), attention masks the keys past a sequence's own length
// (AttnF32Args::seq_len), everything else is row-wise -- so an utterance's mel equals its own single call bit for bit (asserted in
// tests/test_flow_gpu.py).  With U == 1 no length array is passed and the launches are exactly the single-utterance ones.
// finalize = 0 drops the encoder's last pre_lookahead_len * upsample_stride frames (CosyVoice2Model.swift:504-510); enc_chunk / dec_chunk > 0
// switch the conformer encoder / the estimator's transformer blocks to their chunk-masked "streaming" attention
static int flow_inference_core(mia_flow* f, const FlowUtt* utt, int U, int n_timesteps, int finalize, int enc_chunk, int dec_chunk, int mem) {
  if (!f) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = f->ctx;
  const mia_flow_config& c = f->cfg;
  MIA_CHECK_ARG(ctx, utt && U >= 1 && U <= 64 && (mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE), "flow_inference: bad argument (1..64 utterances)");
  const int S = n_timesteps > 0 ? n_timesteps : c.n_timesteps;
  const int st = c.upsample_stride, M = c.output_size, C = c.dec_channels, H = c.dec_heads;
  const int trim = finalize ? 0 : c.pre_lookahead_len * st;
  MIA_CHECK_ARG(ctx, S <= 1000 && enc_chunk >= 0 && dec_chunk >= 0, "flow_inference: at most 1000 steps, chunk sizes >= 0");
  int Tt = 0;                                  // longest token sequence
  std::vector<int32_t> lens((size_t)4 * U);    // [tokens | encoder frames | estimator frames of the cond, uncond sequences]
  std::vector<int> Tu(U);
  for (int u = 0; u < U; ++u) {
    const FlowUtt& q = utt[u];
    MIA_CHECK_ARG(ctx, q.token && q.embedding && q.z && q.mel && q.n_token > 0 && q.n_prompt >= 0 && (q.n_prompt == 0 || q.prompt_token),
                  "flow_inference: utterance %d: bad argument", u);
    const int tt = q.n_token + q.n_prompt, te = tt * st;
    MIA_CHECK_ARG(ctx, tt <= 4096, "flow_inference: at most 4096 tokens");
    Tu[u] = te > trim ? te - trim : te;       // (the reference trims only when something is left, :507)
    MIA_CHECK_ARG(ctx, q.prompt_feat_len >= 0 && q.prompt_feat_len < Tu[u] && (q.prompt_feat_len == 0 || q.prompt_feat),
                  "flow_inference: utterance %d: prompt_feat_len must be in [0, %d)", u, Tu[u]);
    lens[u] = tt; lens[U + u] = te; lens[2 * U + 2 * u] = Tu[u]; lens[2 * U + 2 * u + 1] = Tu[u];
    Tt = std::max(Tt, tt);
  }
  const int T_enc = Tt * st;
  const int T = T_enc > trim ? T_enc - trim : T_enc;      // rows per sequence of the estimator's stack
  MIA_CHECK_ARG(ctx, (int64_t)2 * U * T_enc * std::max(4 * C, 4 * M) < (int64_t)1 << 31, "flow_inference: batch too large");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  Bufs b;
  if (int rc = flow_prepare(f, Tt, S, b, U)) return rc;
  hipStream_t s = ctx->stream;
  const hipMemcpyKind kind = mem == MIA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  int32_t* d_len = f->d_ids + (size_t)U * Tt;
  if (U > 1) {
    MIA_HIP(ctx, hipMemsetAsync(f->d_ids, 0, (size_t)U * Tt * 4, s));     // padding ids: a valid table row
    MIA_HIP(ctx, hipMemcpyAsync(d_len, lens.data(), lens.size() * 4, hipMemcpyHostToDevice, s));
  }
  for (int u = 0; u < U; ++u) {
    const FlowUtt& q = utt[u];
    if (int rc = upload_tokens(f, f->d_ids + (size_t)u * Tt, q.prompt_token, q.n_prompt, q.token, q.n_token, mem)) return rc;
    MIA_HIP(ctx, hipMemcpyAsync(b.emb + (size_t)u * c.spk_embed_dim, q.embedding, (size_t)c.spk_embed_dim * 4, kind, s));
    MIA_HIP(ctx, hipMemcpyAsync(b.z + (size_t)u * T * M, q.z, (size_t)Tu[u] * M * 4, kind, s));
    if (q.prompt_feat_len) MIA_HIP(ctx, hipMemcpyAsync(b.pf + (size_t)u * T * M, q.prompt_feat, (size_t)q.prompt_feat_len * M * 4, kind, s));
  }
  const int32_t* len_tok = U > 1 ? d_len : nullptr;
  const int32_t* len_up = U > 1 ? d_len + U : nullptr;
  const int32_t* len_est = U > 1 ? d_len + 2 * U : nullptr;
  if (int rc = run_encoder(f, b, f->d_ids, Tt, enc_chunk, U, len_tok, len_up)) return rc;
  const unsigned gTM = (unsigned)(((int64_t)T * M + 255) / 256);
  for (int u = 0; u < U; ++u) {
    hipLaunchKernelGGL(flow_spks, dim3(1), dim3(256), 0, s, b.emb + (size_t)u * c.spk_embed_dim, f->spk.w, f->spk.b, b.spks + (size_t)u * M, c.spk_embed_dim, M);
    hipLaunchKernelGGL(flow_pack_inputs, dim3(gTM), dim3(256), 0, s, b.z + (size_t)u * T * M, b.mu + (size_t)u * T_enc * M, b.spks + (size_t)u * M,
                       b.pf + (size_t)u * T * M, b.hin + (size_t)u * 2 * T * 4 * M, b.xs + (size_t)u * T * M, Tu[u], M, utt[u].prompt_feat_len, T);
  }

  // ---- time grid (cosine schedule) and time embeddings of every step, in float32 like the reference (CosyVoice2CFM.swift:89-92,127-181)
  std::vector<float> tspan(S + 1), tval(S), dts(S);
  for (int i = 0; i <= S; ++i) {
    const float lin = S == 0 ? 0.f : (float)((double)i / (double)S);
    tspan[i] = 1.0f - cosf(lin * 0.5f * 3.14159274101257324f);
  }
  {
    float t = tspan[0], dt = tspan[1] - tspan[0];
    for (int k = 1; k <= S; ++k) {
      tval[k - 1] = t; dts[k - 1] = dt;
      t = t + dt;
      if (k < S) dt = tspan[k + 1] - t;
    }
  }
  const int IC = c.dec_in_channels, half = IC / 2, TE = 4 * C;
  std::vector<float> tsin((size_t)S * IC);
  const float emb_scale = logf(10000.0f) / (float)(half - 1);
  for (int k = 0; k < S; ++k)
    for (int i = 0; i < half; ++i) {
      const float arg = 1000.0f * tval[k] * expf((float)i * -emb_scale);
      tsin[(size_t)k * IC + i] = sinf(arg);
      tsin[(size_t)k * IC + half + i] = cosf(arg);
    }
  MIA_HIP(ctx, hipMemcpyAsync(b.tsin, tsin.data(), tsin.size() * 4, hipMemcpyHostToDevice, s));
  MIA_HIP(ctx, hipStreamSynchronize(s));    // tsin / lens are stack-lifetime host buffers
  Run r{f, s};
  r.gemm(f->t1, b.tsin, IC, S, b.te1, TE, 4);
  r.gemm(f->t2, b.te1, TE, S, b.te2, TE);
  hipLaunchKernelGGL(flow_mish, dim3((unsigned)(((int64_t)S * TE + 255) / 256)), dim3(256), 0, s, b.te2, b.tm, (int64_t)S * TE);
  r.gemm(f->tproj, b.tm, TE, S, b.tpr, f->n_res * C);

  const int ns = 2 * U, M2 = ns * T;        // the stack: [cond, uncond] of every utterance, T rows each
  for (int k = 0; k < S && r.rc == MIA_OK; ++k) {
    const float* tvec = b.tpr + (size_t)k * f->n_res * C;
    // down block
    resnet(r, f->down.rn, b, b.hin, IC, T, C, tvec, b.xr, ns);
    for (const TBlock& t : f->down.tb) tblock(r, t, b, b.xr, T, C, H, dec_chunk, ns, len_est);
    // skip -> right half of the up block's input; causal "downsample" conv (stride 1 for the single-level U-Net)
    MIA_HIP(ctx, hipMemcpy2DAsync(b.cat + C, (size_t)2 * C * 4, b.xr, (size_t)C * 4, (size_t)C * 4, M2, hipMemcpyDeviceToDevice, s));
    float* cur = b.xr2; float* nxt = b.xr;
    r.gemm(f->down_conv, b.xr, C, T, cur, C, 0, nullptr, 2, ns);
    for (const UBlock& mb : f->mid) {
      resnet(r, mb.rn, b, cur, C, T, C, tvec, nxt, ns);
      for (const TBlock& t : mb.tb) tblock(r, t, b, nxt, T, C, H, dec_chunk, ns, len_est);
      std::swap(cur, nxt);
    }
    MIA_HIP(ctx, hipMemcpy2DAsync(b.cat, (size_t)2 * C * 4, cur, (size_t)C * 4, (size_t)C * 4, M2, hipMemcpyDeviceToDevice, s));
    resnet(r, f->up.rn, b, b.cat, 2 * C, T, C, tvec, nxt, ns);
    for (const TBlock& t : f->up.tb) tblock(r, t, b, nxt, T, C, H, dec_chunk, ns, len_est);
    r.gemm(f->up_conv2, nxt, C, T, cur, C, 0, nullptr, 2, ns);
    // final block + projection
    r.gemm(f->final_conv, cur, C, T, b.c1, C, 0, nullptr, 2, ns);
    r.ln_mish(f->final_n, b.c1, nullptr, b.h1, M2, C);
    r.gemm(f->final_proj, b.h1, C, M2, b.d, M);
    hipLaunchKernelGGL(flow_euler, dim3(gTM, U), dim3(256), 0, s, b.d, b.xs, b.hin, T, M, dts[k], c.cfg_rate);
  }
  if (r.rc != MIA_OK) return r.rc;
  size_t out_off = 0;
  for (int u = 0; u < U; ++u) {
    const int To = Tu[u] - utt[u].prompt_feat_len;
    if (utt[u].mel_frames) *utt[u].mel_frames = To;
    float* dst = mem == MIA_MEM_DEVICE ? utt[u].mel : b.out + out_off;
    hipLaunchKernelGGL(flow_emit, dim3((unsigned)(((int64_t)To * M + 255) / 256)), dim3(256), 0, s, b.xs + (size_t)u * T * M, dst, Tu[u], M, utt[u].prompt_feat_len);
    if (mem == MIA_MEM_HOST) MIA_HIP(ctx, hipMemcpyAsync(utt[u].mel, dst, (size_t)To * M * 4, hipMemcpyDeviceToHost, s));
    out_off += (size_t)To * M;
  }
  MIA_HIP(ctx, hipGetLastError());
  if (mem == MIA_MEM_HOST) MIA_HIP(ctx, hipStreamSynchronize(s));
  return MIA_OK;
}

static int flow_inference_impl(mia_flow* f, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt, const float* prompt_feat,
                               int prompt_feat_len, const float* embedding, const float* z, int n_timesteps, int finalize, int enc_chunk, int dec_chunk,
                               float* mel, int* mel_frames, int mem) {
  if (!f) return MIA_ERR_MODEL_NOT_LOADED;
  const FlowUtt u{token, n_token, prompt_token, n_prompt, prompt_feat, prompt_feat_len, embedding, z, mel, mel_frames};
  return flow_inference_core(f, &u, 1, n_timesteps, finalize, enc_chunk, dec_chunk, mem);
}

int mia_flow_inference(mia_flow* f, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt, const float* prompt_feat,
                       int prompt_feat_len, const float* embedding, const float* z, int n_timesteps, float* mel, int mem) {
  return flow_inference_impl(f, token, n_token, prompt_token, n_prompt, prompt_feat, prompt_feat_len, embedding, z, n_timesteps, 1, 0, 0, mel, nullptr, mem);
}

int mia_flow_inference_batch(mia_flow* f, int n_utt, const int32_t* const* token, const int32_t* n_token, const int32_t* const* prompt_token,
                             const int32_t* n_prompt, const float* const* prompt_feat, const int32_t* prompt_feat_len, const float* const* embedding,
                             const float* const* z, int n_timesteps, float* const* mel, int mem) {
  if (!f) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_CHECK_ARG(f->ctx, n_utt >= 1 && n_utt <= 64 && token && n_token && n_prompt && prompt_feat_len && embedding && z && mel,
                "flow_inference_batch: bad argument (1..64 utterances)");
  std::vector<FlowUtt> u((size_t)n_utt);
  for (int i = 0; i < n_utt; ++i)
    u[i] = FlowUtt{token[i], n_token[i], prompt_token ? prompt_token[i] : nullptr, n_prompt[i], prompt_feat ? prompt_feat[i] : nullptr,
                   prompt_feat_len[i], embedding[i], z[i], mel[i], nullptr};
  return flow_inference_core(f, u.data(), n_utt, n_timesteps, 1, 0, 0, mem);
}

int mia_flow_inference_streaming(mia_flow* f, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt, const float* prompt_feat,
                                 int prompt_feat_len, const float* embedding, const float* z, int n_timesteps, int finalize, int enc_static_chunk,
                                 int dec_static_chunk, float* mel, int* mel_frames, int mem) {
  return flow_inference_impl(f, token, n_token, prompt_token, n_prompt, prompt_feat, prompt_feat_len, embedding, z, n_timesteps, finalize ? 1 : 0,
                             enc_static_chunk, dec_static_chunk, mel, mel_frames, mem);
}

}  // extern "C"
