// dp.hip -- data-parallel plumbing behind the C ABI (SURVEY.md section 8b "multi-GPU", 8e): clips shard across the GPUs of one
// node with a full weight replica per rank; the path's ONLY exchange is one all-gather of the int32 token rows per pass, over
// RCCL / xGMI on the context's own HIP stream (stream-ordered behind the decode, no host synchronisation).
//
// The reference has no multi-device path (one Swift actor per model, one Metal device: STT/Whisper/WhisperSTT.swift:11); the
// shard rule below is the one bench.py and parallel.py use (contiguous shards, the first n % world ranks one item larger).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy the process already holds when torch is loaded): libmia.so keeps no
// link-time dependency on it, so the CPU-side ABI tests load the library on machines without RCCL, and a missing / broken RCCL is
// reported by mia_dp_init instead of failing the whole library load.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>

#include "mia_internal.h"

namespace {

typedef struct { char internal[128]; } rcclUniqueId;   // ncclUniqueId (rccl.h:43, NCCL_UNIQUE_ID_BYTES 128)
typedef void* rcclComm;
enum { RCCL_INT32 = 2 };                               // ncclInt32 / ncclInt (rccl.h ncclDataType_t)

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm*, int, rcclUniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rcclComm, hipStream_t) = nullptr;
  int (*CommDestroy)(rcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};

void rccl_bind(RcclApi& api) {
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) break;
  }
  if (!api.lib) { api.err = std::string("dlopen(librccl) failed: ") + (dlerror() ? dlerror() : "?"); return; }
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
  api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) { api.err = "librccl lacks the nccl* entry points"; api.lib = nullptr; }
}

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] { rccl_bind(api); });
  return api;
}

int rccl_fail(mia_ctx* ctx, const char* what, int rc) {
  const char* s = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
  return mia_fail(ctx, MIA_ERR_DEVICE, "%s failed: %s (%d)", what, s, rc);
}

// rows [src_row0, +n) of the gathered, padded image -> rows [dst_row0, +n) of the dense result
__global__ void dp_compact_rows(const int32_t* __restrict__ padded, int32_t* __restrict__ dense, const int32_t* __restrict__ plan, int L) {
  // plan[3*r + {0,1,2}] = (first padded row, first dense row, rows) of rank r; blockIdx.y = rank
  const int r = blockIdx.y;
  const int src0 = plan[3 * r], dst0 = plan[3 * r + 1], n = plan[3 * r + 2];
  const int64_t total = (int64_t)n * L;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    dense[(int64_t)dst0 * L + e] = padded[(int64_t)src0 * L + e];
}

}  // namespace

// ---- shard arithmetic (host, no GPU): the rule of parallel.py:shard_range ------------------------------------------------------------------
extern "C" int mia_dp_shard_range(int n_items, int rank, int world, int* lo, int* hi) {
  if (world <= 0 || rank < 0 || rank >= world || n_items < 0 || !lo || !hi) return MIA_ERR_INVALID_ARGUMENT;
  const int q = n_items / world, r = n_items % world;
  *lo = rank * q + std::min(rank, r);
  *hi = *lo + q + (rank < r ? 1 : 0);
  return MIA_OK;
}

extern "C" int mia_dp_shard_cap(int n_items, int world) {
  return world > 0 && n_items >= 0 ? (n_items + world - 1) / world : MIA_ERR_INVALID_ARGUMENT;
}

// The all-gather moves `cap` rows per rank (shards padded to the largest one); this is the unpadding step on HOST buffers, the
// same plan the device path applies with dp_compact_rows.  gathered [world][cap][L] -> dense [n_items][L] in global clip order.
extern "C" int mia_dp_unpack_host(const int32_t* gathered, int n_items, int world, int L, int32_t* dense) {
  if (!gathered || !dense || world <= 0 || n_items < 0 || L <= 0) return MIA_ERR_INVALID_ARGUMENT;
  const int cap = mia_dp_shard_cap(n_items, world);
  for (int r = 0; r < world; ++r) {
    int lo, hi;
    mia_dp_shard_range(n_items, r, world, &lo, &hi);
    if (hi > lo) memcpy(dense + (size_t)lo * L, gathered + (size_t)r * cap * L, (size_t)(hi - lo) * L * sizeof(int32_t));
  }
  return MIA_OK;
}

// ---- RCCL ---------------------------------------------------------------------------------------------------------------------------------
// 1 when RCCL can be bound in this process (dlopen + the four entry points), else 0.  mia_dp_init is a COLLECTIVE: a rank whose
// binding fails would leave its peers blocked inside ncclCommInitRank, so callers vote on this (e.g. an all-reduce MIN over their
// bootstrap channel) before any rank calls mia_dp_init.
extern "C" int mia_dp_available(void) { return rccl().lib ? 1 : 0; }

extern "C" int mia_dp_unique_id(mia_ctx* ctx, void* id128) {
  if (!ctx || !id128) return MIA_ERR_INVALID_ARGUMENT;
  RcclApi& a = rccl();
  if (!a.lib) return mia_fail(ctx, MIA_ERR_DEVICE, "dp_unique_id: %s", a.err.c_str());
  rcclUniqueId id;
  const int rc = a.GetUniqueId(&id);
  if (rc != 0) return rccl_fail(ctx, "ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return MIA_OK;
}

extern "C" int mia_dp_init(mia_ctx* ctx, int rank, int world, const void* unique_id) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, world >= 1 && rank >= 0 && rank < world && unique_id, "dp_init: bad rank/world %d/%d or null id", rank, world);
  MIA_CHECK_ARG(ctx, !ctx->dp_comm, "dp_init: this context already has a communicator (mia_dp_shutdown first)");
  RcclApi& a = rccl();
  if (!a.lib) return mia_fail(ctx, MIA_ERR_DEVICE, "dp_init: %s", a.err.c_str());
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  rcclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  rcclComm comm = nullptr;
  const int rc = a.CommInitRank(&comm, world, id, rank);
  if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
  ctx->dp_comm = comm; ctx->dp_rank = rank; ctx->dp_world = world;
  return MIA_OK;
}

extern "C" int mia_dp_shutdown(mia_ctx* ctx) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  if (ctx->dp_comm) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)rccl().CommDestroy((rcclComm)ctx->dp_comm);
    ctx->dp_comm = nullptr;
  }
  if (ctx->dp_buf) { (void)hipFree(ctx->dp_buf); ctx->dp_buf = nullptr; ctx->dp_buf_bytes = 0; }
  ctx->dp_world = 0; ctx->dp_rank = 0; ctx->dp_plan_items = -1;
  return MIA_OK;
}

// All-gather of this rank's token rows and counts on the context's stream.  local_tokens [b_local][L], local_counts [b_local] and
// the outputs all_tokens [n_items][L], all_counts [n_items] are DEVICE pointers; b_local must be this rank's shard size of n_items.
// Stream-ordered: returns after enqueueing (the caller synchronises the context when it needs the result on the host).
extern "C" int mia_dp_gather_tokens(mia_ctx* ctx, const int32_t* local_tokens, const int32_t* local_counts, int b_local, int L, int n_items,
                                    int32_t* all_tokens, int32_t* all_counts) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, ctx->dp_comm, "dp_gather_tokens: mia_dp_init has not run on this context");
  MIA_CHECK_ARG(ctx, local_tokens && local_counts && all_tokens && all_counts && L > 0 && n_items >= 0, "dp_gather_tokens: null pointer or bad size");
  const int world = ctx->dp_world, rank = ctx->dp_rank;
  int lo, hi;
  mia_dp_shard_range(n_items, rank, world, &lo, &hi);
  MIA_CHECK_ARG(ctx, b_local == hi - lo, "dp_gather_tokens: rank %d holds %d rows, its shard of %d items is %d", rank, b_local, n_items, hi - lo);
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  const int cap = mia_dp_shard_cap(n_items, world);
  if (cap == 0) return MIA_OK;
  // staging: [send: cap rows of (L + 1) ints: tokens | count] [recv: world * cap rows] [plan: 3 * world ints]
  const int W = L + 1;
  const size_t send_b = (size_t)cap * W * 4, recv_b = send_b * world, plan_b = (size_t)3 * world * 4;
  // the unpadding plan sits at a FIXED offset (the start of the buffer): its address must not move with L, because it is uploaded
  // only when (n_items, world) change
  const size_t plan_off = align_up(plan_b, 256);
  const size_t need = plan_off + align_up(send_b, 256) + align_up(recv_b, 256) + align_up(recv_b, 256);
  if (ctx->dp_buf_bytes < need) {
    MIA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dp_buf) (void)hipFree(ctx->dp_buf);
    ctx->dp_buf = nullptr; ctx->dp_buf_bytes = 0;
    if (hipMalloc(&ctx->dp_buf, need) != hipSuccess) return mia_fail(ctx, MIA_ERR_OUT_OF_MEMORY, "dp_gather_tokens: hipMalloc(%zu) failed", need);
    ctx->dp_buf_bytes = need;
    ctx->dp_plan_items = -1;
  }
  char* base = (char*)ctx->dp_buf;
  int32_t* plan = (int32_t*)base;
  int32_t* send = (int32_t*)(base + plan_off);
  int32_t* recv = (int32_t*)(base + plan_off + align_up(send_b, 256));
  int32_t* dense = (int32_t*)(base + plan_off + align_up(send_b, 256) + align_up(recv_b, 256));
  hipStream_t s = ctx->stream;
  // one message per rank: rows of [tokens(L) | count], padded with zero rows up to cap
  MIA_HIP(ctx, hipMemsetAsync(send, 0, send_b, s));
  if (b_local > 0) {
    MIA_HIP(ctx, hipMemcpy2DAsync(send, (size_t)W * 4, local_tokens, (size_t)L * 4, (size_t)L * 4, b_local, hipMemcpyDeviceToDevice, s));
    MIA_HIP(ctx, hipMemcpy2DAsync(send + L, (size_t)W * 4, local_counts, 4, 4, b_local, hipMemcpyDeviceToDevice, s));
  }
  const int rc = rccl().AllGather(send, recv, (size_t)cap * W, RCCL_INT32, (rcclComm)ctx->dp_comm, s);
  if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
  // unpad in global clip order, then split the rows back into tokens and counts.  The plan depends only on (n_items, world):
  // it is uploaded (with one host sync) when that key changes and reused afterwards, so steady-state calls never touch the host.
  int32_t* plan_slot = plan;
  if (ctx->dp_plan_items != n_items) {
    std::vector<int32_t> hplan(3 * world);
    for (int r = 0; r < world; ++r) {
      int l2, h2;
      mia_dp_shard_range(n_items, r, world, &l2, &h2);
      hplan[3 * r] = r * cap; hplan[3 * r + 1] = l2; hplan[3 * r + 2] = h2 - l2;
    }
    MIA_HIP(ctx, hipMemcpyAsync(plan_slot, hplan.data(), hplan.size() * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));   // hplan is a stack-lifetime host buffer
    ctx->dp_plan_items = n_items;
  }
  hipLaunchKernelGGL(dp_compact_rows, dim3(64, world), dim3(256), 0, s, recv, dense, plan_slot, W);
  if (n_items > 0) {
    MIA_HIP(ctx, hipMemcpy2DAsync(all_tokens, (size_t)L * 4, dense, (size_t)W * 4, (size_t)L * 4, n_items, hipMemcpyDeviceToDevice, s));
    MIA_HIP(ctx, hipMemcpy2DAsync(all_counts, 4, dense + L, (size_t)W * 4, 4, n_items, hipMemcpyDeviceToDevice, s));
  }
  MIA_HIP(ctx, hipGetLastError());
  return MIA_OK;
}
