// mia_ctx.hip -- context lifecycle for the C ABI (include/mia.h).
#include <cstdlib>
#include "codec.h"
#include "mia_internal.h"

extern "C" const char* mia_version(void) { return "mia 0.1 (gfx950)"; }

static mia_ctx* create_impl(int device, hipStream_t stream, bool external) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return nullptr;
  if (device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
  // The kernels are compiled for gfx950 only; refuse anything else loudly instead of faulting later.
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fprintf(stderr, "mia: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
    return nullptr;
  }
  mia_ctx* ctx = new mia_ctx();
  ctx->device = device;
  if (external) {
    ctx->stream = stream;
    ctx->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return nullptr;
    }
    ctx->own_stream = true;
  }
  return ctx;
}

extern "C" mia_ctx* mia_create(int device_ordinal) { return create_impl(device_ordinal, nullptr, false); }

extern "C" mia_ctx* mia_create_on_stream(int device_ordinal, void* hip_stream) {
  return create_impl(device_ordinal, (hipStream_t)hip_stream, true);
}

extern "C" void mia_destroy(mia_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  (void)mia_dp_shutdown(ctx);
  for (auto& t : ctx->mel_tables) {
    (void)hipFree(t.window);
    (void)hipFree(t.twiddle);
    (void)hipFree(t.fb_w);
    (void)hipFree(t.fb_meta);
  }
  for (void* p : ctx->table_allocs) (void)hipFree(p);
  if (ctx->s3gen_mel) free(ctx->s3gen_mel);
  mia_resampler_free(ctx);
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* mia_last_error(const mia_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" void* mia_stream(mia_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int mia_synchronize(mia_ctx* ctx) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIA_OK;
}

void* mia_workspace(mia_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return ctx->ws;
  // grow: wait for in-flight users of the old arena first
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  size_t want = align_up(bytes, (size_t)1 << 20);
  if (hipMalloc(&ctx->ws, want) != hipSuccess) {
    mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "workspace hipMalloc(%zu) failed", want);
    return nullptr;
  }
  ctx->ws_bytes = want;
  return ctx->ws;
}

// ---- profiling -------------------------------------------------------------------------------
static hipEvent_t get_event(mia_ctx* ctx) {
  if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

int mia_prof_begin(mia_ctx* ctx, int cls, double work) {
  if (!ctx->prof_on) return -1;
  mia_ctx::ProfRec r{cls, get_event(ctx), get_event(ctx), work};
  if (!r.start || !r.stop) return -1;
  (void)hipEventRecord(r.start, ctx->stream);
  ctx->prof.push_back(r);
  return (int)ctx->prof.size() - 1;
}

void mia_prof_end(mia_ctx* ctx, int rec) {
  if (rec < 0) return;
  (void)hipEventRecord(ctx->prof[rec].stop, ctx->stream);
}

static const char* kProfNames[MIA_PROF_NCLASSES] = {"logmel", "enc_gemm", "enc_attention", "enc_norm", "decode", "crosskv_gemm"};

extern "C" int mia_profile_enable(mia_ctx* ctx, int on) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  ctx->prof_on = on != 0;
  return MIA_OK;
}

extern "C" double mia_profile_codec_bytes(int reset) { return codec_alg_bytes(reset != 0); }

extern "C" int mia_profile_reset(mia_ctx* ctx) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->prof) { ctx->ev_pool.push_back(r.start); ctx->ev_pool.push_back(r.stop); }
  ctx->prof.clear();
  return MIA_OK;
}

extern "C" int mia_profile_read(mia_ctx* ctx, const char* kernel_class, int64_t* launches, double* total_ms, double* total_work) {
  if (!ctx || !kernel_class) return MIA_ERR_INVALID_ARGUMENT;
  int cls = -1;
  for (int i = 0; i < MIA_PROF_NCLASSES; ++i) if (!strcmp(kProfNames[i], kernel_class)) cls = i;
  MIA_CHECK_ARG(ctx, cls >= 0, "profile_read: unknown kernel class '%s'", kernel_class);
  MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int64_t n = 0; double ms = 0.0, work = 0.0;
  for (auto& r : ctx->prof) {
    if (r.cls != cls) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) continue;
    ++n; ms += t; work += r.work;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (total_work) *total_work = work;
  return MIA_OK;
}
