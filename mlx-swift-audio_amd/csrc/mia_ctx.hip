// mia_ctx.hip -- context lifecycle for the C ABI (include/mia.h).
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
  for (auto& t : ctx->mel_tables) {
    (void)hipFree(t.window);
    (void)hipFree(t.twiddle);
    (void)hipFree(t.fb_w);
    (void)hipFree(t.fb_meta);
  }
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
