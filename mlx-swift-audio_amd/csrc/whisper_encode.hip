// whisper_encode.hip -- batched Whisper encoder forward + cross-attention K/V priming on gfx950.
// Replaces AudioEncoder.callAsFunction (STT/Whisper/Layers/AudioEncoder.swift:43-68), the encoder half of
// ResidualAttentionBlock (ResidualAttentionBlock.swift:51-95) and the cached cross K/V projection
// (MultiHeadAttention.swift:49-59).  Launch sequence per layer: LN -> fused QKV GEMM (V transposed in the
// epilogue) -> flash attention -> out-proj GEMM (+bias +residual, fp32) -> LN -> MLP1 GEMM (+bias, erf-GELU)
// -> MLP2 GEMM (+bias +residual).  The two convolutions are GEMMs over overlapping row windows (gemm.hip).
#include <cstdlib>

#include "whisper.h"

namespace {

int gemm(mia_whisper* w, GemmArgs g, int cls = MIA_PROF_ENC_GEMM) {
  g.variant = w->gemm_variant;
  if (const char* e = mia_gemm_check(g)) return mia_fail(w->ctx, MIA_ERR_INVALID_ARGUMENT, "%s", e);
  const int rec = mia_prof_begin(w->ctx, cls, 2.0 * g.M * (double)g.N * g.K * (g.batch > 0 ? g.batch : 1));
  const int rc = mia_gemm_launch(g, w->dtype, w->ctx->stream);
  mia_prof_end(w->ctx, rec);
  if (rc != 0) return mia_fail(w->ctx, MIA_ERR_DEVICE, "gemm launch failed");
  return MIA_OK;
}

int norm(mia_whisper* w, const LNW& ln, void* out, int M, int D) {
  const int rec = mia_prof_begin(w->ctx, MIA_PROF_ENC_NORM, (double)M * D * 6.0);   // fp32 in + 16-bit out
  const int rc = mia_norm_launch(w->x, D, ln.g, ln.b, out, D, M, D, 1e-5f, false, w->dtype, w->ctx->stream);
  mia_prof_end(w->ctx, rec);
  if (rc) return mia_fail(w->ctx, MIA_ERR_DEVICE, "norm launch failed");
  return MIA_OK;
}

}  // namespace

// (Re)allocate every per-batch buffer for capacity B.  Transactional: the new set is allocated into a scratch copy of the handle's
// pointers and committed only when every allocation has succeeded; the superseded set is freed after the stream has drained
// (nothing on the device can still read it) and the captured step graph, which holds the old pointers, is invalidated first.
int whisper_reserve(mia_whisper* w, int B) {
  if (B <= w->cap_B) return MIA_OK;
  MIA_HIP(w->ctx, hipStreamSynchronize(w->ctx->stream));
  w->graph_valid = false;
  const mia_whisper_dims& d = w->dims;
  const size_t D = d.n_audio_state, T = d.n_audio_ctx, H = d.n_audio_head, L = d.n_text_layer;
  const size_t M = (size_t)B * T;
  const int Tpad = (int)align_up(T, 64);
  std::vector<void*> fresh;
  bool ok = true;
  auto get = [&](size_t bytes, bool zero) -> void* {
    if (!ok) return nullptr;
    void* q = nullptr;
    if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) { ok = false; mia_fail(w->ctx, MIA_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed for batch buffers", bytes); return nullptr; }
    fresh.push_back(q);
    if (zero && hipMemsetAsync(q, 0, bytes, w->ctx->stream) != hipSuccess) { ok = false; mia_fail(w->ctx, MIA_ERR_DEVICE, "memset failed"); }
    return q;
  };
  mia_whisper n = *w;   // scratch copy of the pointer set (vectors are copied too; only the raw pointers below are used)
#define A(p, bytes, zero) n.p = (decltype(n.p))get((bytes), (zero))
  A(mel_pad, ((size_t)B * (2 * T + 2) * d.n_mels + 256) * 2, true);
  A(conv1_out, (size_t)B * (2 * T + 1) * D * 2, true);
  A(x, M * D * 4, false);
  A(h, M * D * 2, false);
  A(qk, M * 2 * D * 2, false);
  A(vt, (size_t)B * H * 64 * Tpad * 2, true);
  A(att, M * D * 2, false);
  A(g, M * 4 * D * 2, false);
  A(feat, M * D * 2, false);
  A(enc_part, M * (D / 64 + 1) * 2 * 4, false);
  A(enc_stat, M * 2 * 4, false);
  A(cross_k, L * M * D * 2, false);
  A(cross_v, L * M * D * 2, false);
  // decoder
  const size_t C = d.n_text_ctx, V = d.n_vocab;
  A(self_k, L * B * C * D * 2, true);
  A(self_v, L * B * C * D * 2, true);
  A(dx, (size_t)B * D * 4, false);
  // GEMM operands of the decode step live in MFMA-fragment order (decode.h): whole 32-row blocks, rows past B stay zero
  const size_t B32 = align_up((size_t)B, 32);
  A(dh, B32 * D * 2, true);
  A(dq, (size_t)B * D * 2, false);
  A(da, B32 * D * 2, true);
  A(dg, B32 * 4 * D * 2, true);
  A(partial, (size_t)16 * B * D * 4, false);
  A(dstat, (size_t)2 * ((D + 15) / 16) * B * 2 * 4, true);
  A(logits, (size_t)B * V * 4, false);
  A(tokens, (size_t)B * C * 4, true);
  A(n_gen, (size_t)B * 4, true);
  A(finished, (size_t)B * 4, true);
  A(last_ts, (size_t)B * 4, true);
  A(out_n, (size_t)B * 4, true);
  A(sum_logprob, (size_t)B * 4, true);
  A(n_logprob, (size_t)B * 4, true);
  A(no_speech, (size_t)B * 4, true);
  A(uniforms, (size_t)B * C * 4, true);
  A(out_tokens, (size_t)B * C * 4, true);
  A(out_avg, (size_t)B * 4, true);
  A(clip.pos, (size_t)B * 4, true);
  A(clip.n_init, (size_t)B * 4, true);
  A(clip.sot_idx, (size_t)B * 4, true);
  A(clip.temp, (size_t)B * 4, true);
#undef A
  if (ok && hipStreamSynchronize(w->ctx->stream) != hipSuccess) { ok = false; mia_fail(w->ctx, MIA_ERR_DEVICE, "stream sync failed"); }
  if (!ok) {
    for (void* q : fresh) (void)hipFree(q);
    return w->ctx->err.find("hipMalloc") != std::string::npos ? MIA_ERR_OUT_OF_MEMORY : MIA_ERR_DEVICE;
  }
  if (!w->suppress_bits) {   // vocabulary-sized, independent of B: allocated once, lives with the handle
    void* q = nullptr;
    const size_t bytes = 2 * ((V + 31) / 32) * 4;
    if (hipMalloc(&q, bytes) != hipSuccess) { for (void* f : fresh) (void)hipFree(f); return mia_fail(w->ctx, MIA_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", bytes); }
    (void)hipMemsetAsync(q, 0, bytes, w->ctx->stream);
    w->allocs.push_back(q);
    w->suppress_bits = (uint32_t*)q;
  }
  // commit: raw pointers from the scratch copy, then retire the superseded set (the stream is idle)
  w->mel_pad = n.mel_pad; w->conv1_out = n.conv1_out; w->x = n.x; w->h = n.h; w->qk = n.qk; w->vt = n.vt; w->att = n.att; w->g = n.g;
  w->feat = n.feat; w->enc_part = n.enc_part; w->enc_stat = n.enc_stat; w->cross_k = n.cross_k; w->cross_v = n.cross_v; w->self_k = n.self_k; w->self_v = n.self_v; w->dx = n.dx; w->dh = n.dh;
  w->dq = n.dq; w->da = n.da; w->dg = n.dg; w->partial = n.partial; w->dstat = n.dstat; w->logits = n.logits; w->tokens = n.tokens; w->n_gen = n.n_gen;
  w->finished = n.finished; w->last_ts = n.last_ts; w->out_n = n.out_n; w->sum_logprob = n.sum_logprob; w->n_logprob = n.n_logprob;
  w->no_speech = n.no_speech; w->uniforms = n.uniforms; w->out_tokens = n.out_tokens; w->out_avg = n.out_avg; w->clip = n.clip;
  for (void* q : w->batch_allocs) (void)hipFree(q);
  w->batch_allocs.swap(fresh);
  w->Tpad = Tpad;
  w->cap_B = B;
  return MIA_OK;
}

int whisper_encode_from_padded_mel(mia_whisper* w, int B) {
  const mia_whisper_dims& d = w->dims;
  const int D = d.n_audio_state, T = d.n_audio_ctx, H = d.n_audio_head, M = B * T;
  hipStream_t s = w->ctx->stream;
  int rc;
  {  // conv1 + GELU: rows are overlapping windows of the zero-row-padded mel (stride n_mels, K = 3*n_mels)
    GemmArgs g;
    g.A = w->mel_pad; g.lda = d.n_mels; g.strideA = (int64_t)(2 * T + 2) * d.n_mels;
    g.W = w->conv1.w; g.bias = w->conv1.b; g.act = MIA_ACT_GELU;
    g.C = (uint16_t*)w->conv1_out + D; g.ldc = D; g.strideC = (int64_t)(2 * T + 1) * D;
    g.M = 2 * T; g.N = D; g.K = w->kpad_conv1; g.batch = B;
    if ((rc = gemm(w, g)) != MIA_OK) return rc;
  }
  {  // conv2 (stride 2) + GELU + positional embedding -> fp32 residual stream
    GemmArgs g;
    g.A = w->conv1_out; g.lda = 2 * D; g.strideA = (int64_t)(2 * T + 1) * D;
    g.W = w->conv2.w; g.bias = w->conv2.b; g.act = MIA_ACT_GELU;
    g.R = w->enc_pos; g.ldr = D; g.strideR = 0;
    g.C = w->x; g.ldc = D; g.strideC = (int64_t)T * D; g.out_f32 = 1;
    g.M = T; g.N = D; g.K = 3 * D; g.batch = B;
    if ((rc = gemm(w, g)) != MIA_OK) return rc;
  }
  for (int l = 0; l < d.n_audio_layer; ++l) {
    const EncBlockW& b = w->enc[l];
    if ((rc = norm(w, b.attn_ln, w->h, M, D)) != MIA_OK) return rc;
    {
      GemmArgs g;
      g.A = w->h; g.lda = D; g.W = b.qkv.w; g.bias = b.qkv.b;
      g.C = w->qk; g.ldc = 2 * D; g.C2 = w->vt;
      g.M = M; g.N = 3 * D; g.K = D; g.epi = MIA_EPI_QKV_VT; g.T = T; g.H = H; g.Tpad = w->Tpad;
      if ((rc = gemm(w, g)) != MIA_OK) return rc;
    }
    if (const char* e = mia_enc_attention_check(B, T, H, w->Tpad, 2 * D, D)) return mia_fail(w->ctx, MIA_ERR_INVALID_ARGUMENT, "%s", e);
    {
      const int rec = mia_prof_begin(w->ctx, MIA_PROF_ENC_ATTN, 4.0 * B * H * (double)T * T * 64);
      const int arc = mia_enc_attention_launch(w->qk, 2 * D, w->vt, w->att, D, B, T, H, w->Tpad, w->dtype, s);
      mia_prof_end(w->ctx, rec);
      if (arc) return mia_fail(w->ctx, MIA_ERR_DEVICE, "attention launch failed");
    }
    {
      // out-proj (+ residual) and the MLP's first Linear; when both run on the 8-phase kernel's row epilogues the LayerNorm between them
      // is carried through the GEMMs (gemm.h) instead of a pass of its own over the fp32 stream
      GemmArgs go, g1;
      go.A = w->att; go.lda = D; go.W = b.out.w; go.bias = b.out.b;
      go.R = w->x; go.ldr = D; go.C = w->x; go.ldc = D; go.out_f32 = 1;
      go.M = M; go.N = D; go.K = D; go.variant = w->gemm_variant;
      g1.A = w->h; g1.lda = D; g1.W = b.mlp1.w; g1.bias = b.mlp1.b; g1.act = MIA_ACT_GELU;
      g1.C = w->g; g1.ldc = 4 * D; g1.M = M; g1.N = 4 * D; g1.K = D; g1.variant = w->gemm_variant;
      GemmArgs fo = go, f1 = g1;
      fo.ln_gamma = b.mlp_ln.g; fo.ln_out = w->h; fo.ln_ld = D; fo.ln_part = w->enc_part;
      f1.ln_stat = w->enc_stat; f1.ln_c1 = b.mlp1.c1; f1.bias = b.mlp1.c2;
      const bool carry = b.mlp1.c1 && b.mlp1.c2 && mia_gemm_ln_ok(fo) && mia_gemm_ln_ok(f1);
      if ((rc = gemm(w, carry ? fo : go)) != MIA_OK) return rc;
      if (carry) {
        const int rec = mia_prof_begin(w->ctx, MIA_PROF_ENC_NORM, (double)M * (D / 64) * 8.0);
        const int frc = mia_ln_finalize_launch(w->enc_part, D / 64, D, 1e-5f, w->enc_stat, M, s);
        mia_prof_end(w->ctx, rec);
        if (frc) return mia_fail(w->ctx, MIA_ERR_DEVICE, "ln finalize launch failed");
      } else if ((rc = norm(w, b.mlp_ln, w->h, M, D)) != MIA_OK) return rc;
      if ((rc = gemm(w, carry ? f1 : g1)) != MIA_OK) return rc;
    }
    {
      GemmArgs g;
      g.A = w->g; g.lda = 4 * D; g.W = b.mlp2.w; g.bias = b.mlp2.b;
      g.R = w->x; g.ldr = D; g.C = w->x; g.ldc = D; g.out_f32 = 1;
      g.M = M; g.N = D; g.K = 4 * D;
      if ((rc = gemm(w, g)) != MIA_OK) return rc;
    }
  }
  if ((rc = norm(w, w->ln_post, w->feat, M, D)) != MIA_OK) return rc;
  // cross-attention K/V of every decoder layer, head-major [L][B][H][T][64]
  for (int l = 0; l < d.n_text_layer; ++l) {
    const DecBlockW& b = w->dec[l];
    const size_t off = (size_t)l * w->cap_B * T * D;   // layer stride uses the capacity so buffers stay put when B varies
    for (int kv = 0; kv < 2; ++kv) {
      GemmArgs g;
      g.A = w->feat; g.lda = D; g.W = kv ? b.cv.w : b.ck.w; g.bias = kv ? b.cv.b : b.ck.b;
      g.C = (uint16_t*)(kv ? w->cross_v : w->cross_k) + off;
      g.M = M; g.N = D; g.K = D; g.epi = MIA_EPI_HEADMAJOR; g.T = T; g.H = H;
      if ((rc = gemm(w, g, MIA_PROF_CROSSKV_GEMM)) != MIA_OK) return rc;
    }
  }
  w->cur_B = B;
  MIA_HIP(w->ctx, hipGetLastError());
  return MIA_OK;
}

// dense [B][2T][n_mels] (compute dtype) -> zero-row-padded layout
static __global__ void pad_mel_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t rows, int n_mels, int B) {
  const int64_t per = rows * n_mels;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < per * B; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / per, r = e - b * per;
    out[b * (rows + 2) * n_mels + n_mels + r] = in[e];
  }
}

extern "C" int mia_whisper_encode(mia_whisper* w, const void* mel, int B, int mem) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_CHECK_ARG(ctx, mel && B > 0, "whisper_encode: mel must be non-null and B > 0");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "whisper_encode: bad mem %d", mem);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = whisper_reserve(w, B);
  if (rc != MIA_OK) return rc;
  const int64_t rows = 2 * (int64_t)w->dims.n_audio_ctx;
  const size_t bytes = (size_t)B * rows * w->dims.n_mels * 2;
  const uint16_t* src = (const uint16_t*)mel;
  if (mem == MIA_MEM_HOST) {
    void* ws = mia_workspace(ctx, bytes);
    if (!ws) return MIA_ERR_OUT_OF_MEMORY;
    MIA_HIP(ctx, hipMemcpyAsync(ws, mel, bytes, hipMemcpyHostToDevice, ctx->stream));
    src = (const uint16_t*)ws;
  }
  hipLaunchKernelGGL(pad_mel_kernel, dim3(1024), dim3(256), 0, ctx->stream, src, (uint16_t*)w->mel_pad, rows, w->dims.n_mels, B);
  return whisper_encode_from_padded_mel(w, B);
}

template <typename T>
static __global__ void cvt16_to_f32(const uint16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) out[e] = T::to_f32(in[e]);
}

extern "C" int mia_whisper_get_audio_features(mia_whisper* w, void* out, int dtype, int mem) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  mia_ctx* ctx = w->ctx;
  MIA_CHECK_ARG(ctx, out && w->cur_B > 0, "get_audio_features: no encode has run (or null out)");
  MIA_CHECK_ARG(ctx, dtype == MIA_F32 || dtype == w->dtype, "get_audio_features: dtype must be MIA_F32 or the compute dtype");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t n = (int64_t)w->cur_B * w->dims.n_audio_ctx * w->dims.n_audio_state;
  const void* src = w->feat;
  size_t bytes = (size_t)n * 2;
  if (dtype == MIA_F32) {
    bytes = (size_t)n * 4;
    float* dst = mem == MIA_MEM_DEVICE ? (float*)out : (float*)mia_workspace(ctx, bytes);
    if (!dst) return MIA_ERR_OUT_OF_MEMORY;
    if (w->dtype == MIA_F16) hipLaunchKernelGGL(cvt16_to_f32<F16>, dim3(1024), dim3(256), 0, ctx->stream, (const uint16_t*)w->feat, dst, n);
    else hipLaunchKernelGGL(cvt16_to_f32<BF16>, dim3(1024), dim3(256), 0, ctx->stream, (const uint16_t*)w->feat, dst, n);
    if (mem == MIA_MEM_DEVICE) return MIA_OK;
    src = dst;
  }
  MIA_HIP(ctx, hipMemcpyAsync(out, src, bytes, mem == MIA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  if (mem == MIA_MEM_HOST) MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIA_OK;
}

// Test hook (tests/test_whisper_gpu.py): force the encoder GEMM tile variant so that every kernel is exercised at reduced sizes.
// 0 / 1: 128^2 tile (register / LDS-DMA staged), 2: 256^2 two-buffer, 3: auto (default), 4: 256^2 8-phase.
extern "C" int mia_whisper_set_gemm_variant(mia_whisper* w, int variant) {
  if (!w) return MIA_ERR_MODEL_NOT_LOADED;
  MIA_CHECK_ARG(w->ctx, variant >= 0 && variant <= 4, "set_gemm_variant: variant must be 0..4 (got %d)", variant);
  w->gemm_variant = variant;
  return MIA_OK;
}
