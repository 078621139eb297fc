// whisper_load.hip -- upload a Whisper checkpoint (named host tensors) into the HBM layout of whisper.h.
// Replaces the tensor side of WhisperModel.load (STT/Whisper/WhisperModel.swift:184-206); key names are the
// reference's Module property paths (Layers/*.swift @ModuleInfo keys).
#include <cmath>

#include "decode.h"

namespace {

inline float half_to_float(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1f, m = h & 0x3ff;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = s << 31;
    else {
      int sh = 0; uint32_t mm = m;
      while (!(mm & 0x400)) { mm <<= 1; ++sh; }
      u = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3ff) << 13);
    }
  } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
  else u = (s << 31) | ((e - 15 + 127) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
inline uint16_t float_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
inline uint16_t float_to_half(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                      // overflow -> inf (after rounding)
  if (x < 0x38800000u) {                                                         // subnormal / zero
    if (x < 0x33000000u) return (uint16_t)sign;
    const int shift = 113 - (int)(x >> 23);
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    const uint32_t round_bit = 1u << (shift + 12);
    uint32_t r = m >> (shift + 13);
    const uint32_t rem = m & ((round_bit << 1) - 1);
    if (rem > round_bit || (rem == round_bit && (r & 1))) ++r;
    return (uint16_t)(sign | r);
  }
  uint32_t r = (x - 0x38000000u) >> 13;
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
  return (uint16_t)(sign | r);
}

struct Loader {
  mia_whisper* w;
  std::map<std::string, const mia_tensor_view*> by_name;
  std::string err;

  const mia_tensor_view* find(const std::string& n, bool required = true) {
    auto it = by_name.find(n);
    if (it == by_name.end()) {
      if (required && err.empty()) err = "missing tensor '" + n + "'";
      return nullptr;
    }
    return it->second;
  }
  static int64_t numel(const mia_tensor_view* t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
  }
  // read element i of a host tensor as fp32
  static void to_f32(const mia_tensor_view* t, std::vector<float>& out) {
    const int64_t n = numel(t);
    out.resize(n);
    if (t->dtype == MIA_F32) memcpy(out.data(), t->data, n * 4);
    else if (t->dtype == MIA_F16) { const uint16_t* p = (const uint16_t*)t->data; for (int64_t i = 0; i < n; ++i) out[i] = half_to_float(p[i]); }
    else { const uint16_t* p = (const uint16_t*)t->data; for (int64_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)p[i] << 16; memcpy(&out[i], &u, 4); } }
  }
  void* dev_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { if (err.empty()) err = "hipMalloc failed (weights)"; return nullptr; }
    w->allocs.push_back(p);
    return p;
  }
  float* upload_f32(const std::vector<float>& v) {
    float* d = (float*)dev_alloc(v.size() * 4);
    if (d && hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    return d;
  }
  void* upload_16(const std::vector<float>& v) {
    std::vector<uint16_t> q(v.size());
    if (w->dtype == MIA_F16) for (size_t i = 0; i < v.size(); ++i) q[i] = float_to_half(v[i]);
    else for (size_t i = 0; i < v.size(); ++i) q[i] = float_to_bf16(v[i]);
    void* d = dev_alloc(q.size() * 2);
    if (d && hipMemcpy(d, q.data(), q.size() * 2, hipMemcpyHostToDevice) != hipSuccess && err.empty()) err = "hipMemcpy failed";
    return d;
  }
  bool expect(const mia_tensor_view* t, std::initializer_list<int64_t> shp, const std::string& name) {
    if (!t) return false;
    bool ok = t->ndim == (int)shp.size();
    int i = 0;
    for (int64_t s : shp) { if (ok && t->shape[i] != s) ok = false; ++i; }
    if (!ok && err.empty()) err = "tensor '" + name + "' has an unexpected shape";
    return ok;
  }
  LNW ln(const std::string& p, int D) {
    LNW o;
    const mia_tensor_view* g = find(p + ".weight"); const mia_tensor_view* b = find(p + ".bias");
    if (!expect(g, {D}, p + ".weight") || !expect(b, {D}, p + ".bias")) return o;
    std::vector<float> v; to_f32(g, v); o.g = upload_f32(v); to_f32(b, v); o.b = upload_f32(v);
    return o;
  }
  // one or several [N_i][K] matrices stacked along N; bias_flags[i] says whether part i has a bias
  LinearW linear(const std::vector<std::string>& parts, const std::vector<bool>& has_bias, int N_each, int K) {
    LinearW o; o.N = N_each * (int)parts.size(); o.K = K;
    std::vector<float> wv((size_t)o.N * K), bv((size_t)o.N, 0.f), tmp;
    bool any_bias = false;
    for (size_t i = 0; i < parts.size(); ++i) {
      const mia_tensor_view* t = find(parts[i] + ".weight");
      if (!expect(t, {N_each, K}, parts[i] + ".weight")) return o;
      to_f32(t, tmp); memcpy(&wv[i * (size_t)N_each * K], tmp.data(), tmp.size() * 4);
      if (has_bias[i]) {
        const mia_tensor_view* b = find(parts[i] + ".bias");
        if (!expect(b, {N_each}, parts[i] + ".bias")) return o;
        to_f32(b, tmp); memcpy(&bv[i * (size_t)N_each], tmp.data(), tmp.size() * 4); any_bias = true;
      }
    }
    o.w = upload_16(wv);
    if (any_bias) o.b = upload_f32(bv);
    return o;
  }
  // Conv1d weight [Cout][3][Cin] -> [Cout][Kpad] (tap-major rows == the overlapping-window GEMM's K order)
  LinearW conv(const std::string& p, int Cout, int Cin, int Kpad) {
    LinearW o; o.N = Cout; o.K = Kpad;
    const mia_tensor_view* t = find(p + ".weight"); const mia_tensor_view* b = find(p + ".bias");
    if (!expect(t, {Cout, 3, Cin}, p + ".weight") || !expect(b, {Cout}, p + ".bias")) return o;
    std::vector<float> src, wv((size_t)Cout * Kpad, 0.f), bv;
    to_f32(t, src);
    for (int c = 0; c < Cout; ++c) memcpy(&wv[(size_t)c * Kpad], &src[(size_t)c * 3 * Cin], (size_t)3 * Cin * 4);
    o.w = upload_16(wv);
    to_f32(b, bv); o.b = upload_f32(bv);
    return o;
  }
};

}  // namespace

extern "C" mia_whisper* mia_whisper_load(mia_ctx* ctx, const mia_whisper_dims* dims, const mia_tensor_view* tensors,
                                         int n_tensors, int compute_dtype) {
  if (!ctx) return nullptr;
  auto fail = [&](const std::string& m) -> mia_whisper* { ctx->err = "whisper_load: " + m; return nullptr; };
  if (!dims || !tensors || n_tensors <= 0) return fail("null arguments");
  if (compute_dtype != MIA_BF16 && compute_dtype != MIA_F16) return fail("compute_dtype must be MIA_BF16 or MIA_F16");
  const mia_whisper_dims& d = *dims;
  if (d.n_audio_state <= 0 || d.n_audio_state % 64 || d.n_audio_head * 64 != d.n_audio_state)
    return fail("n_audio_state must equal 64 * n_audio_head (Whisper head dim is 64)");
  if (d.n_text_state % 64 || d.n_text_head * 64 != d.n_text_state) return fail("n_text_state must equal 64 * n_text_head");
  if (d.n_text_state != d.n_audio_state) return fail("n_text_state must equal n_audio_state");
  if (d.n_text_state > 2048) return fail("n_text_state must be <= 2048 (decode row kernels keep a row in registers)");
  if (d.n_mels <= 0 || d.n_mels > 128 || d.n_mels % 8) return fail("n_mels must be a multiple of 8 in 8..128");
  if (d.n_audio_ctx <= 0 || d.n_audio_ctx % 4) return fail("n_audio_ctx must be a positive multiple of 4");
  if (d.n_vocab <= 0 || d.n_text_ctx <= 0 || d.n_audio_layer <= 0 || d.n_text_layer <= 0) return fail("bad dims");
  if (hipSetDevice(ctx->device) != hipSuccess) return fail("hipSetDevice failed");

  mia_whisper* w = new mia_whisper();
  w->ctx = ctx; w->dims = d; w->dtype = compute_dtype;
  w->kpad_conv1 = (int)align_up((size_t)3 * d.n_mels, 64);
  Loader L; L.w = w;
  for (int i = 0; i < n_tensors; ++i) {
    if (!tensors[i].name || !tensors[i].data) { mia_whisper_free(w); return fail("tensor view with null name/data"); }
    L.by_name[tensors[i].name] = &tensors[i];
  }
  const int D = d.n_audio_state;
  w->conv1 = L.conv("encoder.conv1", D, d.n_mels, w->kpad_conv1);
  w->conv2 = L.conv("encoder.conv2", D, D, 3 * D);
  {
    std::vector<float> pos;
    if (const mia_tensor_view* t = L.find("encoder.positional_embedding", false)) {
      if (L.expect(t, {d.n_audio_ctx, D}, "encoder.positional_embedding")) Loader::to_f32(t, pos);
    } else {  // sinusoids(length:channels:) AudioEncoder.swift:78-96, fp32 math
      pos.resize((size_t)d.n_audio_ctx * D);
      const float inc = logf(10000.0f) / (float)(D / 2 - 1);
      for (int p = 0; p < d.n_audio_ctx; ++p)
        for (int i = 0; i < D / 2; ++i) {
          const float st = (float)p * expf(-inc * (float)i);
          pos[(size_t)p * D + i] = sinf(st);
          pos[(size_t)p * D + D / 2 + i] = cosf(st);
        }
    }
    if (!pos.empty()) w->enc_pos = L.upload_f32(pos);
  }
  w->enc.resize(d.n_audio_layer);
  for (int l = 0; l < d.n_audio_layer && L.err.empty(); ++l) {
    const std::string p = "encoder.blocks." + std::to_string(l);
    EncBlockW& b = w->enc[l];
    b.attn_ln = L.ln(p + ".attn_ln", D);
    b.qkv = L.linear({p + ".attn.query", p + ".attn.key", p + ".attn.value"}, {true, false, true}, D, D);
    b.out = L.linear({p + ".attn.out"}, {true}, D, D);
    b.mlp_ln = L.ln(p + ".mlp_ln", D);
    b.mlp1 = L.linear({p + ".mlp1"}, {true}, 4 * D, D);
    b.mlp2 = L.linear({p + ".mlp2"}, {true}, D, 4 * D);
  }
  w->ln_post = L.ln("encoder.ln_post", D);
  if (const mia_tensor_view* t = L.find("decoder.token_embedding.weight")) {
    if (L.expect(t, {d.n_vocab, D}, "decoder.token_embedding.weight")) { std::vector<float> v; Loader::to_f32(t, v); w->tok_emb = L.upload_16(v); }
  }
  if (const mia_tensor_view* t = L.find("decoder.positional_embedding")) {
    if (L.expect(t, {d.n_text_ctx, D}, "decoder.positional_embedding")) { std::vector<float> v; Loader::to_f32(t, v); w->dec_pos = L.upload_f32(v); }
  }
  w->dec.resize(d.n_text_layer);
  for (int l = 0; l < d.n_text_layer && L.err.empty(); ++l) {
    const std::string p = "decoder.blocks." + std::to_string(l);
    DecBlockW& b = w->dec[l];
    b.attn_ln = L.ln(p + ".attn_ln", D);
    b.qkv = L.linear({p + ".attn.query", p + ".attn.key", p + ".attn.value"}, {true, false, true}, D, D);
    b.out = L.linear({p + ".attn.out"}, {true}, D, D);
    b.cross_ln = L.ln(p + ".cross_attn_ln", D);
    b.cq = L.linear({p + ".cross_attn.query"}, {true}, D, D);
    b.ck = L.linear({p + ".cross_attn.key"}, {false}, D, D);
    b.cv = L.linear({p + ".cross_attn.value"}, {true}, D, D);
    b.cout = L.linear({p + ".cross_attn.out"}, {true}, D, D);
    b.mlp_ln = L.ln(p + ".mlp_ln", D);
    b.mlp1 = L.linear({p + ".mlp1"}, {true}, 4 * D, D);
    b.mlp2 = L.linear({p + ".mlp2"}, {true}, D, 4 * D);
  }
  w->dec_ln = L.ln("decoder.ln", D);
  if (!L.err.empty()) { mia_whisper_free(w); return fail(L.err); }
  {  // the decode step reads its weights in MFMA-fragment order (decode.h): one repack per matrix, on the device
    bool ok = true;
    auto frag = [&](const void* src, int N, int K) -> void* {
      if (!ok || !src) return nullptr;
      void* dst = nullptr;
      const size_t bytes = (size_t)((N + 15) / 16) * 16 * K * 2;
      if (K % 32 != 0 || hipMalloc(&dst, bytes) != hipSuccess) { ok = false; return nullptr; }
      w->allocs.push_back(dst);
      if (dec_launch_repack_wfrag(src, dst, N, K, ctx->stream) != 0) ok = false;
      return dst;
    };
    for (DecBlockW& b : w->dec)
      for (LinearW* lw : {&b.qkv, &b.out, &b.cq, &b.cout, &b.mlp1, &b.mlp2}) lw->wf = frag(lw->w, lw->N, lw->K);
    w->tok_emb_f = frag(w->tok_emb, d.n_vocab, D);
    if (!ok) { mia_whisper_free(w); return fail("fragment-order repack of the decoder weights failed"); }
    // LayerNorm fold constants of every decoder Linear that consumes a LayerNorm (whisper.h LinearW::c1 / c2; decode.h)
    auto fold = [&](const void* w16, int N, const LNW& ln, float** c1, float** c2, const float* bias = nullptr) {
      if (!ok || !w16) return;
      void* p1 = nullptr; void* p2 = nullptr;
      if (hipMalloc(&p1, (size_t)N * 4) != hipSuccess) { ok = false; return; }
      w->allocs.push_back(p1);
      if (hipMalloc(&p2, (size_t)N * 4) != hipSuccess) { ok = false; return; }
      w->allocs.push_back(p2);
      *c1 = (float*)p1; *c2 = (float*)p2;
      if (dec_launch_lnfold(w16, N, D, ln.g, ln.b, *c1, *c2, w->dtype, ctx->stream, bias) != 0) ok = false;
    };
    for (DecBlockW& b : w->dec) {
      fold(b.qkv.w, b.qkv.N, b.attn_ln, &b.qkv.c1, &b.qkv.c2);
      fold(b.cq.w, b.cq.N, b.cross_ln, &b.cq.c1, &b.cq.c2);
      fold(b.mlp1.w, b.mlp1.N, b.mlp_ln, &b.mlp1.c1, &b.mlp1.c2);
    }
    fold(w->tok_emb, d.n_vocab, w->dec_ln, &w->emb_c1, &w->emb_c2);
    // encoder: the Linear behind mlp_ln takes its LayerNorm through the GEMM (gemm.h "LayerNorm carried across two GEMMs"); c2 includes the bias
    if (d.n_audio_state == D)
      for (EncBlockW& b : w->enc) fold(b.mlp1.w, b.mlp1.N, b.mlp_ln, &b.mlp1.c1, &b.mlp1.c2, b.mlp1.b);
    if (!ok) { mia_whisper_free(w); return fail("LayerNorm fold of the decoder weights failed"); }
  }
  if (hipDeviceSynchronize() != hipSuccess) { mia_whisper_free(w); return fail("device error during upload"); }
  return w;
}

extern "C" void mia_whisper_free(mia_whisper* w) {
  if (!w) return;
  if (w->n_clones > 0) { mia_fail(w->ctx, MIA_ERR_INVALID_ARGUMENT, "whisper_free: %d clone(s) still share these weights; free them first", w->n_clones); return; }
  (void)hipSetDevice(w->ctx->device);
  (void)hipStreamSynchronize(w->ctx->stream);
  if (w->step_graph) (void)hipGraphExecDestroy(w->step_graph);
  if (w->step_graph_n) (void)hipGraphExecDestroy(w->step_graph_n);
  for (void* p : w->batch_allocs) (void)hipFree(p);
  for (void* p : w->allocs) (void)hipFree(p);
  if (w->ev_enc_begin) (void)hipEventDestroy(w->ev_enc_begin);
  if (w->ev_enc_end) (void)hipEventDestroy(w->ev_enc_end);
  if (w->trace) (void)hipFree(w->trace);
  if (w->trace_clips) (void)hipFree(w->trace_clips);
  if (w->parent) w->parent->n_clones -= 1;
  delete w;
}

// A second handle on the SAME weights with its own activations, KV caches, decode state and step graph, bound to another context
// (HIP stream) of the same device: batches decoded on different streams then stream one weight copy (the second reader mostly hits
// L2 / the Infinity Cache) instead of one copy each.
extern "C" mia_whisper* mia_whisper_clone(mia_whisper* src, mia_ctx* ctx) {
  if (!src || !ctx) return nullptr;
  mia_whisper* root = src->parent ? src->parent : src;
  if (ctx->device != root->ctx->device) { mia_fail(ctx, MIA_ERR_INVALID_ARGUMENT, "whisper_clone: the context must be on the device that holds the weights"); return nullptr; }
  mia_whisper* w = new mia_whisper();
  w->ctx = ctx; w->dims = root->dims; w->dtype = root->dtype; w->kpad_conv1 = root->kpad_conv1;
  w->conv1 = root->conv1; w->conv2 = root->conv2; w->enc_pos = root->enc_pos; w->enc = root->enc; w->ln_post = root->ln_post;
  w->tok_emb = root->tok_emb; w->tok_emb_f = root->tok_emb_f; w->emb_c1 = root->emb_c1; w->emb_c2 = root->emb_c2; w->dec_pos = root->dec_pos; w->dec = root->dec; w->dec_ln = root->dec_ln;
  w->parent = root;
  root->n_clones += 1;
  return w;
}
