// mia_internal.h -- shared host-side plumbing for the HIP layer (context, workspace, error handling).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mia.h"

struct mia_resampler_cache;

struct mia_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // grow-only scratch arena in HBM (never freed between calls: no hipMalloc on the hot path)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // cached per-n_mels front-end tables (device)
  struct MelTables {
    int n_mels = 0;
    int window_kind = -1;
    float* window = nullptr;    // [400]
    float* twiddle = nullptr;   // [400][2][224] cos|sin
    float* fb_w = nullptr;      // compact non-zero filter weights
    int* fb_meta = nullptr;     // [n_mels][3] = (first bin, count, offset into fb_w)
    int fb_nnz = 0;
  };
  std::vector<MelTables> mel_tables;
  std::vector<void*> table_allocs;   // other cached device tables (freed at destroy), e.g. the 24 kHz 80-mel front end
  void* s3gen_mel = nullptr;         // S3GenMelTables* (mel_s3gen.hip), lives in table_allocs' lifetime
  mia_resampler_cache* resampler = nullptr;   // per-ratio polyphase filters of mia_resample_sinc (resample.hip); device tables in table_allocs
  // optional HIP-event profiling of kernel classes (mia_profile_*): bench.py's roofline figures come from here
  bool prof_on = false;
  struct ProfRec { int cls; hipEvent_t start, stop; double work; };
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  // data-parallel exchange (dp.hip): one RCCL communicator per context, collectives run on the context's stream
  void* dp_comm = nullptr;
  int dp_rank = 0, dp_world = 0, dp_plan_items = -1;
  void* dp_buf = nullptr;
  size_t dp_buf_bytes = 0;
};

enum { MIA_PROF_LOGMEL = 0, MIA_PROF_ENC_GEMM = 1, MIA_PROF_ENC_ATTN = 2, MIA_PROF_ENC_NORM = 3, MIA_PROF_DECODE = 4,
       MIA_PROF_CROSSKV_GEMM = 5, MIA_PROF_NCLASSES = 6 };

// RAII-less helpers: if profiling is on, bracket the launches between begin/end with events on the ctx stream
int mia_prof_begin(mia_ctx* ctx, int cls, double work);   // returns record index or -1
void mia_prof_end(mia_ctx* ctx, int rec);

inline int mia_fail(mia_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define MIA_HIP(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return mia_fail((ctx), MIA_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                      __FILE__, __LINE__);                                                      \
  } while (0)

#define MIA_CHECK_ARG(ctx, cond, ...)                                   \
  do {                                                                  \
    if (!(cond)) return mia_fail((ctx), MIA_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

// Ensure the ctx workspace holds at least `bytes`; returns nullptr on failure (ctx->err set).
void* mia_workspace(mia_ctx* ctx, size_t bytes);

inline size_t mia_dtype_size(int dt) { return dt == MIA_F32 ? 4 : 2; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

void mia_resampler_free(mia_ctx* ctx);   // resample.hip
