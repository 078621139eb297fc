// dequant.hip -- MLX affine 4 / 8-bit (group 64) de-quantisation of checkpoint tensors at load time.
//
// The reference's default checkpoints are quantised (`quantize(model:) { (64, bits, .affine) }`, STT/Whisper/WhisperModel.swift:189-196;
// TTS/Orpheus/TTSEngine/OrpheusWeightLoader.swift): a Linear / Embedding stores `weight` as uint32 words holding 32/bits codes each
// (code j of a word in bits [j*bits, (j+1)*bits), little end first, along the input axis), plus per-group `scales` and `biases`
// (group = 64 consecutive inputs):  w[o][i] = scales[o][i/64] * code[o][i] + biases[o][i/64].
// MLX evaluates this inside its quantised matmul; here the weights are expanded ONCE to the 16-bit (or fp32) storage type the GEMMs
// stream -- HBM-bound, one pass: 0.5 / 1 byte read and 2 bytes written per weight.
#include "mia_device.h"
#include "mia_internal.h"

namespace {

__device__ __forceinline__ float load_scale(const void* p, int64_t i, int dt) {
  if (dt == MIA_F32) return reinterpret_cast<const float*>(p)[i];
  const uint16_t v = reinterpret_cast<const uint16_t*>(p)[i];
  return dt == MIA_F16 ? F16::to_f32(v) : BF16::to_f32(v);
}

// one thread per uint32 word: 8 (4-bit) or 4 (8-bit) outputs, all in one group (64 % (32/bits) == 0)
template <int BITS>
__global__ __launch_bounds__(256) void dequant_affine_kernel(const uint32_t* __restrict__ wq, const void* __restrict__ scales,
                                                             const void* __restrict__ biases, int64_t rows, int64_t cols, int group,
                                                             int sdt, void* __restrict__ out, int odt) {
  constexpr int PER = 32 / BITS;
  const int64_t words_per_row = cols / PER;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * words_per_row) return;
  const int64_t r = e / words_per_row, wi = e % words_per_row;
  const uint32_t word = wq[e];
  const int64_t c0 = wi * PER;
  const int64_t g = r * (cols / group) + c0 / group;
  const float sc = load_scale(scales, g, sdt), bs = load_scale(biases, g, sdt);
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) v[j] = fmaf(sc, (float)((word >> (j * BITS)) & ((1u << BITS) - 1u)), bs);
  const int64_t o = r * cols + c0;
  if (odt == MIA_F32) {
#pragma unroll
    for (int j = 0; j < PER; ++j) reinterpret_cast<float*>(out)[o + j] = v[j];
  } else {
    uint16_t* op = reinterpret_cast<uint16_t*>(out) + o;
#pragma unroll
    for (int j = 0; j < PER; ++j) op[j] = odt == MIA_F16 ? F16::from_f32(v[j]) : BF16::from_f32(v[j]);
  }
}

}  // namespace

extern "C" int mia_dequant_affine(mia_ctx* ctx, const uint32_t* wq, const void* scales, const void* biases, int64_t rows, int64_t cols,
                                  int group_size, int bits, int scale_dtype, void* out, int out_dtype, int mem) {
  if (!ctx) return MIA_ERR_INVALID_ARGUMENT;
  MIA_CHECK_ARG(ctx, wq && scales && biases && out && rows > 0 && cols > 0, "dequant_affine: null / empty argument");
  MIA_CHECK_ARG(ctx, bits == 4 || bits == 8, "dequant_affine: bits must be 4 or 8");
  MIA_CHECK_ARG(ctx, group_size > 0 && group_size % (32 / bits) == 0 && cols % group_size == 0, "dequant_affine: cols must be a multiple of group_size");
  MIA_CHECK_ARG(ctx, (scale_dtype == MIA_F32 || scale_dtype == MIA_F16 || scale_dtype == MIA_BF16) &&
                         (out_dtype == MIA_F32 || out_dtype == MIA_F16 || out_dtype == MIA_BF16), "dequant_affine: bad dtype");
  MIA_CHECK_ARG(ctx, mem == MIA_MEM_HOST || mem == MIA_MEM_DEVICE, "dequant_affine: bad mem");
  MIA_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t words = rows * cols * bits / 32, groups = rows * (cols / group_size);
  const size_t ssz = scale_dtype == MIA_F32 ? 4 : 2, osz = out_dtype == MIA_F32 ? 4 : 2;
  const uint32_t* d_w = wq; const void* d_s = scales; const void* d_b = biases; void* d_o = out;
  if (mem == MIA_MEM_HOST) {
    auto al = [](size_t n) { return (n + 255) / 256 * 256; };
    const size_t o_s = al((size_t)words * 4), o_b = o_s + al(groups * ssz), o_o = o_b + al(groups * ssz);
    char* ws = (char*)mia_workspace(ctx, o_o + al((size_t)rows * cols * osz));
    if (!ws) return MIA_ERR_OUT_OF_MEMORY;
    MIA_HIP(ctx, hipMemcpyAsync(ws, wq, (size_t)words * 4, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(ws + o_s, scales, groups * ssz, hipMemcpyHostToDevice, s));
    MIA_HIP(ctx, hipMemcpyAsync(ws + o_b, biases, groups * ssz, hipMemcpyHostToDevice, s));
    d_w = (const uint32_t*)ws; d_s = ws + o_s; d_b = ws + o_b; d_o = ws + o_o;
  }
  const unsigned grid = (unsigned)((words + 255) / 256);
  if (bits == 4) hipLaunchKernelGGL(dequant_affine_kernel<4>, dim3(grid), dim3(256), 0, s, d_w, d_s, d_b, rows, cols, group_size, scale_dtype, d_o, out_dtype);
  else hipLaunchKernelGGL(dequant_affine_kernel<8>, dim3(grid), dim3(256), 0, s, d_w, d_s, d_b, rows, cols, group_size, scale_dtype, d_o, out_dtype);
  MIA_HIP(ctx, hipGetLastError());
  if (mem == MIA_MEM_HOST) {
    MIA_HIP(ctx, hipMemcpyAsync(out, d_o, (size_t)rows * cols * osz, hipMemcpyDeviceToHost, s));
    MIA_HIP(ctx, hipStreamSynchronize(s));
  }
  return MIA_OK;
}
