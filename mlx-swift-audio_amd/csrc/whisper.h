// whisper.h -- device-resident Whisper model + per-batch state (internal).
//
// HBM layout (sized for 288 GB: every buffer is allocated once per batch capacity, nothing is re-allocated or
// concatenated per step, unlike the reference's concat-grown KV cache, MultiHeadAttention.swift:67-68):
//   weights      16-bit (bf16|f16) [out][in] row-major; q|k|v of each attention fused into one [3D][D] matrix
//   LN params, biases, positional tables: fp32
//   residual stream x: fp32 [B*T][D];   GEMM operands: 16-bit
//   encoder V:     [B][H][64][Tpad] (transposed per head, zero padded) written by the QKV GEMM epilogue
//   cross K / V:   [L][B][H][T_audio][64] head-major, written once per clip by the encode call
//   self  K / V:   [L][B][H][n_text_ctx][64] head-major ring, written in place at position `pos`
#pragma once
#include <map>
#include <string>
#include <vector>

#include "gemm.h"
#include "mia_device.h"
#include "mia_internal.h"
#include "ops.h"

struct LinearW {
  void* w = nullptr;       // 16-bit [N][K]
  void* wf = nullptr;      // decoder only: the same matrix in MFMA-fragment order (decode.h), read by the step's skinny GEMMs
  float* b = nullptr;      // fp32 [N] or null
  // decoder Linears fed by a LayerNorm: c1[n] = sum_k W[n][k] gamma[k], c2[n] = sum_k W[n][k] beta[k] (fp32, of the 16-bit weights), so
  // that LN(x) W^T = rstd (W (x * gamma) - mean c1) + c2 and the step can feed x * gamma (decode.h, "LayerNorm carried across the chain")
  float* c1 = nullptr;
  float* c2 = nullptr;
  int N = 0, K = 0;
};

struct LNW {
  float* g = nullptr;
  float* b = nullptr;
};

struct EncBlockW {
  LNW attn_ln, mlp_ln;
  LinearW qkv, out, mlp1, mlp2;
};

struct DecBlockW {
  LNW attn_ln, cross_ln, mlp_ln;
  LinearW qkv, out;             // self attention (fused q|k|v)
  LinearW cq, ck, cv, cout;     // cross attention
  LinearW mlp1, mlp2;
};

// Per-clip decode state, device arrays [B] read by every kernel of a step (so ONE captured hipGraph replays for every step
// and every call).  Clips of a batch may sit at different positions: forced prefixes (prompt conditioning) differ in length.
struct DecClip {
  int32_t* pos;       // tokens already in the self-KV cache == position of the token being consumed
  int32_t* n_init;    // length of the forced prefix ([sot_prev]+prompt+sot sequence)
  int32_t* sot_idx;   // position of <|startoftranscript|> (no-speech probe)
  float* temp;        // sampling temperature (0 = greedy argmax)
};

struct DecodeParams {  // immutable per graph (kernel argument, by value)
  int B, V, D, H, L, n_ctx;     // n_ctx = n_text_ctx (448)
  int eot, no_speech, no_timestamps, timestamp_begin;
  int timestamps, max_tokens, max_initial_ts, max_new_tokens;
  int greedy;                   // every clip decodes at temperature 0: the argmax head may be split over several workgroups
  int trace;                    // test hook (mia_whisper_trace_logits): the step also copies the traced clips' logits out
  int head_single;              // test hook (mia_whisper_set_debug bit 1): one-workgroup head even at temperature 0
};

struct mia_whisper {
  mia_ctx* ctx = nullptr;
  mia_whisper_dims dims{};
  int dtype = MIA_BF16;
  int kpad_conv1 = 0;
  int gemm_variant = 3;               // encoder GEMM tile selection (3 = auto); changed only by mia_whisper_set_gemm_variant (tests)
  std::vector<void*> allocs;          // everything hipMalloc'ed for this handle (a clone owns only its batch buffers)
  std::vector<void*> batch_allocs;    // the per-batch buffers of the current capacity (replaced as a set by whisper_reserve)
  mia_whisper* parent = nullptr;      // clone: the handle whose weights are shared (read-only)
  int n_clones = 0;                   // live clones of this handle: it cannot be freed before them

  // ---- weights
  LinearW conv1, conv2;
  float* enc_pos = nullptr;           // [n_audio_ctx][D]
  std::vector<EncBlockW> enc;
  LNW ln_post;
  void* tok_emb = nullptr;            // 16-bit [V][D]  (row gather for the embedding)
  void* tok_emb_f = nullptr;          // the same in MFMA-fragment order (the logits GEMM of the decode step)
  float* emb_c1 = nullptr;            // fp32 [V]: the token embedding folded with decoder.ln (LinearW::c1 / c2 of the logits GEMM)
  float* emb_c2 = nullptr;
  float* dec_pos = nullptr;           // [n_text_ctx][D]
  std::vector<DecBlockW> dec;
  LNW dec_ln;

  // ---- per-batch activations (capacity cap_B)
  int cap_B = 0, cur_B = 0;
  int Tpad = 0;
  void* mel_pad = nullptr;            // 16-bit [B][2T+2][n_mels] (+slack), rows 0 and 2T+1 zero
  void* conv1_out = nullptr;          // 16-bit [B][2T+1][D], row 0 zero
  float* x = nullptr;                 // fp32 [B*T][D]
  void* h = nullptr;                  // 16-bit [B*T][D]
  void* qk = nullptr;                 // 16-bit [B*T][2D]
  void* vt = nullptr;                 // 16-bit [B][H][64][Tpad]
  void* att = nullptr;                // 16-bit [B*T][D]
  void* g = nullptr;                  // 16-bit [B*T][4D]
  void* feat = nullptr;               // 16-bit [B*T][D]  (audio features == encoder output)
  void* cross_k = nullptr;            // 16-bit [L][B][H][T][64]
  void* cross_v = nullptr;

  // ---- decoder state (capacity cap_B)
  void* self_k = nullptr;             // 16-bit [L][B][H][n_ctx][64]
  void* self_v = nullptr;
  float* dx = nullptr;                // fp32 [B][D] residual
  void* dh = nullptr;                 // 16-bit [B][D]  LN output / GEMM operand
  void* dq = nullptr;                 // 16-bit [B][D]
  void* da = nullptr;                 // 16-bit [B][D]  attention output
  void* dg = nullptr;                 // 16-bit [B][4D]
  float* partial = nullptr;           // fp32 [S_max][B][D] split-K partials
  int weight_sharing = 0;             // mia_whisper_set_weight_sharing: other handles stream the same weights concurrently
  float* enc_part = nullptr;          // encoder LayerNorm hand-over (gemm.h): [B T][D / 64][2] partial sums, and
  float* enc_stat = nullptr;          // [B T][2] (mean, rstd)
  float* dstat = nullptr;             // fp32 [2][D / 16][B][2]: per-tile (sum x, sum x^2) of the residual rows, written by the SK_RESID projections
  float* logits = nullptr;            // fp32 [B][V]
  int32_t* tokens = nullptr;          // int32 [B][n_ctx]  full sequence (initial + generated)
  int32_t* n_gen = nullptr;           // int32 [B]  generated count (incl. a trailing EOT while decoding)
  int32_t* finished = nullptr;        // int32 [B]
  int32_t* last_ts = nullptr;         // int32 [B] last generated timestamp token (> timestamp_begin), 0 = none
  int32_t* out_n = nullptr;           // int32 [B]
  float* sum_logprob = nullptr;       // fp32 [B]
  int32_t* n_logprob = nullptr;       // int32 [B]
  float* no_speech = nullptr;         // fp32 [B]
  uint32_t* suppress_bits = nullptr;  // [2][ceil(V/32)]  base mask, base+first-step mask
  float* uniforms = nullptr;          // fp32 [B][n_ctx]
  DecClip clip{};
  int32_t* out_tokens = nullptr;      // int32 [B][n_ctx] compacted outputs
  float* out_avg = nullptr;           // fp32 [B]
  hipGraphExec_t step_graph = nullptr;
  hipGraphExec_t step_graph_n = nullptr;   // DEC_GRAPH_STEPS consecutive steps in one graph (one replay gap instead of DEC_GRAPH_STEPS)
  DecodeParams graph_params{};
  bool graph_valid = false;

  // ---- optional second stream for the encoder half (mia_whisper_set_encode_stream): the log-mel + encoder of a pass is enqueued there,
  // ordered against the decode stream by events, so that a host can give the (latency-bound) decode chain a higher stream priority
  // than the (throughput-bound) encoder of another batch running beside it
  hipStream_t enc_stream = nullptr;
  hipEvent_t ev_enc_begin = nullptr, ev_enc_end = nullptr;

  // ---- test hooks (never set by the product path)
  int debug_flags = 0;                // mia_whisper_set_debug: bit 0 = launch every step directly (no hipGraph), bit 1 = one-workgroup head
  float* trace = nullptr;             // mia_whisper_trace_logits: fp32 [trace_n][n_text_ctx][V], row p = the logits computed at position p
  int32_t* trace_clips = nullptr;     // device int32 [trace_n]: batch rows traced
  int trace_n = 0;
  std::vector<int32_t> trace_clip_ids;
};

// whisper_encode.hip
int whisper_reserve(mia_whisper* w, int B);
int whisper_encode_from_padded_mel(mia_whisper* w, int B);  // mel already in w->mel_pad
// whisper_decode.hip
int whisper_decode(mia_whisper* w, const mia_decode_opts* o, int32_t* tokens, int32_t* n_tokens, float* avg_logprob,
                   float* no_speech_prob, int mem);
// logmel.hip
size_t mia_logmel_scratch_bytes(int B, int64_t n_out, int n_mels);
int mia_logmel_device(mia_ctx* ctx, const float* pcm_dev, const int64_t* offs_host, int B, int n_mels, int window_kind,
                      int64_t pad_right, int64_t n_out, void* out_dev, int out_dtype, bool channel_major,
                      int64_t clip_stride, int64_t row_stride, int64_t col_stride, int64_t row_off, void* scratch);
