// ops_abi.hip -- operator-level C ABI entry points (include/mia.h "operator level").
#include "gemm.h"
#include "mia_internal.h"
#include "ops.h"

extern "C" int mia_op_linear(mia_ctx* ctx, const void* x, int64_t lda, const void* w, const float* bias, const float* r, int64_t ldr,
                             void* y, int64_t ldy, int M, int N, int K, int act, int dtype, int out_f32, int variant, int mem) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, x && w && y, "op_linear: null pointer");
  MIA_CHECK_ARG(ctx, dtype == MIA_BF16 || dtype == MIA_F16, "op_linear: dtype must be MIA_BF16 or MIA_F16");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "op_linear: bad mem");
  MIA_CHECK_ARG(ctx, M > 0 && N > 0 && K > 0 && lda >= K && ldy >= N && (!r || ldr >= N), "op_linear: bad shape");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldy; g.ldr = ldr; g.act = act; g.out_f32 = out_f32 ? 1 : 0; g.variant = variant;
  const size_t xb = (size_t)M * lda * 2, wb = (size_t)N * K * 2, yb = (size_t)M * ldy * (out_f32 ? 4 : 2);
  const size_t bb = bias ? (size_t)N * 4 : 0, rb = r ? (size_t)M * ldr * 4 : 0;
  if (mem == MIA_MEM_DEVICE) {
    g.A = x; g.W = w; g.C = y; g.bias = bias; g.R = r;
  } else {
    const size_t o_w = align_up(xb, 256), o_y = o_w + align_up(wb, 256), o_b = o_y + align_up(yb, 256), o_r = o_b + align_up(bb, 256);
    char* ws = (char*)mia_workspace(ctx, o_r + align_up(rb, 256));
    if (!ws) return MIA_ERR_OUT_OF_MEMORY;
    MIA_HIP(ctx, hipMemcpyAsync(ws, x, xb, hipMemcpyHostToDevice, ctx->stream));
    MIA_HIP(ctx, hipMemcpyAsync(ws + o_w, w, wb, hipMemcpyHostToDevice, ctx->stream));
    if (bias) MIA_HIP(ctx, hipMemcpyAsync(ws + o_b, bias, bb, hipMemcpyHostToDevice, ctx->stream));
    if (r) MIA_HIP(ctx, hipMemcpyAsync(ws + o_r, r, rb, hipMemcpyHostToDevice, ctx->stream));
    g.A = ws; g.W = ws + o_w; g.C = ws + o_y; g.bias = bias ? (const float*)(ws + o_b) : nullptr; g.R = r ? (const float*)(ws + o_r) : nullptr;
  }
  if (const char* e = mia_gemm_check(g)) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "%s", e);
  if (mia_gemm_launch(g, dtype, ctx->stream) != 0) return mia_fail(ctx, MIA_ERR_DEVICE, "op_linear: launch failed");
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(y, g.C, yb, hipMemcpyDeviceToHost, ctx->stream));
    MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MIA_OK;
}

// fp32 Conv1d / Linear on the exact-fp32 matrix cores (device pointers only): y[t][n] = act(b[n] + sum_{k,c} x[t*stride + k*dil - pad][c] w[n][k][c]) (+ r)
#include "codec.h"
extern "C" int mia_op_conv1d_f32(mia_ctx* ctx, const float* x, int64_t ldx, int T_in, const float* w, const float* bias, const float* r, float* y,
                                 int64_t ldy, int T_out, int N, int Cin, int taps, int stride, int dil, int pad, int act) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, x && w && y && T_in > 0 && T_out > 0 && N > 0 && taps > 0 && stride > 0 && dil > 0, "op_conv1d_f32: bad argument");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  ConvGemmArgs g;
  g.X = x; g.ldx = ldx; g.T_in = T_in; g.W = w; g.bias = bias; g.Y = y; g.ldy = ldy; g.T_out = T_out; g.R = r; g.ldr = ldy;
  g.M = T_out; g.N = N; g.Cin = Cin; g.taps = taps; g.dil = dil; g.pad = pad; g.x_row_mul = stride; g.gelu = act;
  if (const char* e = codec_conv_gemm_check(g)) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "%s", e);
  if (codec_conv_gemm_launch(g, 1, ctx->stream)) return mia_fail(ctx, MIA_ERR_DEVICE, "op_conv1d_f32: launch failed");
  return MIA_OK;
}

// fp32 scaled-dot-product attention, head dim 64, full (unmasked) softmax; device pointers (MLXFast.scaledDotProductAttention as used by
// Codec/S3Gen/Matcha/MatchaTransformer.swift:58-66): q/k/v [B*T][ld] with head h in columns h*64.., out [B*T][ldo].
extern "C" int mia_op_attention_f32(mia_ctx* ctx, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out,
                                    int64_t ldo, int B, int T, int H, float scale) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  AttnF32Args a;
  a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo; a.B = B; a.T = T; a.H = H; a.scale = scale;
  if (const char* e = mia_attn_f32_check(a)) return mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "%s", e);
  if (mia_attn_f32_launch(a, ctx->stream)) return mia_fail(ctx, MIA_ERR_DEVICE, "op_attention_f32: launch failed");
  return MIA_OK;
}
