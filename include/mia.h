/* mia.h -- C ABI of the MI355X-native speech-inference hot path ("mia" = MI355X audio).
 *
 * This is the drop-in boundary described in SURVEY.md section 8(b).  The reference
 * (smdesai/mlx-swift-audio, 100 % Swift on MLX) has no FFI seam of its own; its model actors call
 * MLXArray ops directly.  Every entry point below replaces one such call site and cites it
 * (paths relative to /root/reference/package).  A Swift binding is shown in INTEGRATION.md.
 *
 * Conventions (mirroring the actor model of the reference, WhisperSTT.swift:11):
 *   - opaque handles; one mia_ctx == one device + one HIP stream; calls on a ctx are serialised by
 *     the caller (like the Swift actor); no globals; no exceptions cross the ABI.
 *   - every function returns mia_status (0 = ok, <0 = error mirroring STTError cases,
 *     Models/STTError.swift:6-46); mia_last_error(ctx) returns a human readable string.
 *   - caller owns all input/output buffers.  `mem` says where they live: MIA_MEM_HOST (plain host
 *     pointers, the library copies) or MIA_MEM_DEVICE (HBM pointers valid on the ctx device; no copy,
 *     no synchronisation -- work is merely enqueued on the ctx stream).
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point fails with
 *     MIA_ERR_DEVICE.
 */
#ifndef MIA_H
#define MIA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mia_ctx mia_ctx;
typedef struct mia_whisper mia_whisper;
typedef struct mia_codec mia_codec;
typedef struct mia_lm mia_lm;
typedef struct mia_s3tok mia_s3tok;

typedef enum {
  MIA_OK = 0,
  MIA_ERR_INVALID_ARGUMENT = -1, /* STTError.invalidArgument */
  MIA_ERR_MODEL_NOT_LOADED = -2, /* STTError.modelNotLoaded */
  MIA_ERR_INVALID_AUDIO = -3,    /* STTError.invalidAudio */
  MIA_ERR_OUT_OF_MEMORY = -4,
  MIA_ERR_DEVICE = -5,           /* no usable gfx950 device / HIP failure */
  MIA_ERR_UNSUPPORTED = -6
} mia_status;

typedef enum { MIA_F32 = 0, MIA_F16 = 1, MIA_BF16 = 2, MIA_U32 = 3 /* packed quantisation codes */ } mia_dtype;
typedef enum { MIA_MEM_HOST = 0, MIA_MEM_DEVICE = 1 } mia_mem;

/* ---- context -------------------------------------------------------------------------------- */
/* Library version string (static storage). */
const char* mia_version(void);
/* Create a context on `device_ordinal` with a private non-blocking HIP stream. NULL on failure. */
mia_ctx* mia_create(int device_ordinal);
/* Same, but enqueue on a caller-provided hipStream_t (e.g. torch's current stream). */
mia_ctx* mia_create_on_stream(int device_ordinal, void* hip_stream);
void mia_destroy(mia_ctx* ctx);
const char* mia_last_error(const mia_ctx* ctx);
/* hipStream_t of the context (for events recorded by the caller). */
void* mia_stream(mia_ctx* ctx);
/* Block until everything enqueued on the ctx stream has finished. */
int mia_synchronize(mia_ctx* ctx);

/* ---- profiling ------------------------------------------------------------------------------ */
/* Optional HIP-event timing of kernel classes on the ctx stream (events bracket each launch of the class).
 * Classes: "logmel" (work = algorithmic bytes), "enc_gemm", "crosskv_gemm", "enc_attention" (work = FLOPs),
 * "enc_norm" (bytes), "decode" (work = decoder steps).  Reading synchronises the stream. */
int mia_profile_enable(mia_ctx* ctx, int on);
int mia_profile_reset(mia_ctx* ctx);
int mia_profile_read(mia_ctx* ctx, const char* kernel_class, int64_t* launches, double* total_ms, double* total_work);
/* Algorithmic bytes of the fp32 codec / flow / vocoder launches (tap GEMMs, depthwise convs, embeds, noise blocks) issued by the CALLING
 * HOST THREAD since the last reset: per launch each distinct input row once + each output element once + the weights once + residual
 * inputs once (SURVEY.md 8(d)'s byte model for SNAC / DAC / HiFT).  reset != 0 clears the counter after reading. */
double mia_profile_codec_bytes(int reset);

/* ---- DSP front end -------------------------------------------------------------------------- */
/* Whisper log-mel.  Replaces whisperLogMelSpectrogram(audio:nMels:padding:)
 * (STT/Whisper/WhisperAudio.swift:78-137, called at STT/Whisper/WhisperSTT.swift:140-145) and the
 * shared stft/melFilters it uses (Codec/S3Tokenizer/S3TokenizerUtils.swift:224-375).
 *   pcm      float32 mono 16 kHz, clips concatenated; clip b = pcm[offs[b] .. offs[b+1])
 *   offs     int64[B+1], ALWAYS a host pointer
 *   n_mels   80 or 128
 *   pad_right zero samples appended to every clip before the STFT (reference passes 480000)
 *   n_frames_out frames emitted per clip; frame f >= (len_b+pad_right)/160 is written as 0.0
 *            (padOrTrimMel, WhisperSTT.swift:624-635).  The per-clip global max of the max-8 clamp
 *            always covers the whole padded utterance, as in the reference.
 *   mel      out, [B][n_frames_out][n_mels] time-major, dtype out_dtype
 */
int mia_logmel_whisper(mia_ctx* ctx, const float* pcm, const int64_t* offs, int B, int n_mels,
                       int64_t pad_right, int64_t n_frames_out, void* mel, int out_dtype, int mem);

/* S3Tokenizer log-mel (periodic Hann, channel-major output [B][n_mels][n_frames_out]).  Replaces
 * logMelSpectrogram / logMelSpectrogramChatterbox (Codec/S3Tokenizer/S3TokenizerUtils.swift:102-208,
 * called at TTS/CosyVoice2/CosyVoice2TTS.swift:386). */
int mia_logmel_s3(mia_ctx* ctx, const float* pcm, const int64_t* offs, int B, int n_mels,
                  int64_t pad_right, int64_t n_frames_out, void* mel, int out_dtype, int mem);

/* 80-bin 24 kHz log-mel of the CosyVoice2 / S3Gen prompt features.  Replaces s3genMelSpectrogram
 * (Codec/S3Gen/Mel/S3GenMel.swift:43-102, called from TTS/CosyVoice2/CosyVoice2TTS.swift:370-430): reflect pad 720,
 * n_fft 1920, hop 480, periodic Hann, |rfft|, slaney filterbank 0..8 kHz, natural log clamped at 1e-5.
 *   pcm float32 mono 24 kHz [n_samples]; mel out float32 [80][mia_mel_s3gen_frames(n_samples)] (channel-major). */
int64_t mia_mel_s3gen_frames(int64_t n_samples);
int mia_mel_s3gen(mia_ctx* ctx, const float* pcm, int64_t n_samples, float* mel, int mem);

/* Linear-interpolation resampler of the CosyVoice2 prompt path (resampleAudio -> linearInterpolate1d, TTS/CosyVoice2/CosyVoice2TTS.swift:733-744,
 * TTS/CosyVoice2/HiFiGAN/CosyHiFTGenerator.swift:17-60; align_corners = false index rule, float32 arithmetic as in the reference).
 * scale = to_rate / from_rate as float32; out holds mia_resample_linear_len(n_samples, scale) samples. */
int64_t mia_resample_linear_len(int64_t n_samples, float scale);
int mia_resample_linear(mia_ctx* ctx, const float* x, int64_t n_samples, float scale, float* out, int mem);
/* Anti-aliased sample-rate conversion of input audio (e.g. 44.1 / 48 kHz files -> 16 kHz for Whisper, -> 24 kHz for the codecs).  The
 * reference uses AVAudioConverter here (Audio/AudioResampler.swift:15-88), whose filter is not specified in its sources: this is a
 * documented polyphase windowed-sinc interpolator (Kaiser beta 8.6, 16 zero crossings per side, roll-off 0.945, unit DC gain per
 * phase; definition in csrc/resample.hip) -- a stated substitute, not a parity claim.  n_out = floor(n_samples * to / from)
 * (mia_resample_sinc_len).  mia_resample_sinc_table returns the phase filters (float32 [L][taps], L = to / gcd) for inspection. */
int64_t mia_resample_sinc_len(int64_t n_samples, int from_rate, int to_rate);
int mia_resample_sinc_table(int from_rate, int to_rate, float* table, int capacity, int* L_out, int* taps_out);
int mia_resample_sinc(mia_ctx* ctx, const float* x, int64_t n_samples, int from_rate, int to_rate, float* out, int64_t out_capacity,
                      int64_t* n_out, int mem);

/* ---- data-parallel exchange (SURVEY.md section 8b "multi-GPU", 8e) ------------------------------------------------- */
/* Clips shard across the GPUs of one node (one process and one mia_ctx per GPU, full weight replica per rank); the path's only
 * exchange is ONE all-gather of the int32 token rows per pass, over RCCL / xGMI, enqueued on the context's own stream.  The
 * reference has no multi-device path (one actor, one Metal device: STT/Whisper/WhisperSTT.swift:11); these entry points are what
 * a Swift host driving 8 processes would bind.
 *   shard rule: contiguous shards, the first n_items % world ranks hold one item more (mia_dp_shard_range, host arithmetic);
 *   mia_dp_available: 1 when RCCL can be bound in this process; mia_dp_init is a collective, so ranks vote on this first;
 *   mia_dp_unique_id: rank 0 makes the 128-byte RCCL id and hands it to the other ranks by its own means (pipe, file, env);
 *   mia_dp_init: collective over all ranks (ncclCommInitRank); one communicator per context; mia_dp_shutdown destroys it
 *     (mia_destroy does too);
 *   mia_dp_gather_tokens: DEVICE pointers; local_tokens [b_local][L] + local_counts [b_local] of this rank ->
 *     all_tokens [n_items][L] + all_counts [n_items] in global clip order on EVERY rank; stream-ordered, no host sync in steady state;
 *   mia_dp_unpack_host: the unpadding step of the gather on host buffers (gathered [world][cap][L], cap = mia_dp_shard_cap). */
int mia_dp_shard_range(int n_items, int rank, int world, int* lo, int* hi);
int mia_dp_shard_cap(int n_items, int world);
int mia_dp_unpack_host(const int32_t* gathered, int n_items, int world, int L, int32_t* dense);
int mia_dp_available(void);
int mia_dp_unique_id(mia_ctx* ctx, void* id128);
int mia_dp_init(mia_ctx* ctx, int rank, int world, const void* unique_id);
int mia_dp_shutdown(mia_ctx* ctx);
int mia_dp_gather_tokens(mia_ctx* ctx, const int32_t* local_tokens, const int32_t* local_counts, int b_local, int L, int n_items,
                         int32_t* all_tokens, int32_t* all_counts);

/* ---- operator level -------------------------------------------------------------------------- */
/* y = act(x W^T + b) + r : the dense contraction behind every MLXNN Linear on the path
 * (e.g. STT/Whisper/Layers/MultiHeadAttention.swift:40-58,134; ResidualAttentionBlock.swift:91).
 *   x [M][lda] and w [N][K] in `dtype` (MIA_BF16|MIA_F16), y [M][ldy] in `dtype` or fp32 (out_f32),
 *   bias fp32 [N] or NULL, r fp32 [M][ldr] or NULL, act 0 = none / 1 = exact-erf GELU.  K % 64 == 0.
 *   variant 0 = register-staged tiles, 1 = LDS-DMA staged tiles (default elsewhere).  Buffers live in `mem`. */
int mia_op_linear(mia_ctx* ctx, const void* x, int64_t lda, const void* w, const float* bias, const float* r, int64_t ldr,
                  void* y, int64_t ldy, int M, int N, int K, int act, int dtype, int out_f32, int variant, int mem);

/* fp32 Conv1d / Linear, the contraction behind every MLXNN Conv1d / Linear of the codec, flow and vocoder stages
 * (e.g. Codec/S3Gen/S3GenDecoder.swift:40-48, Codec/S3Gen/Matcha/MatchaTransformer.swift:36-56), time-major, DEVICE pointers:
 *   y[t][n] = act(b[n] + sum_{k,c} x[t*stride + k*dil - pad][c] * w[n][k][c]) (+ r[t][n]);  x [T_in][ldx], w [N][taps][Cin] (Cin % 32 == 0),
 *   act 0 none / 1 exact-erf GELU / 2 ELU / 3 abs / 4 SiLU / 5 leaky-ReLU(0.01).  Rows outside [0, T_in) read as zero. */
int mia_op_conv1d_f32(mia_ctx* ctx, const float* x, int64_t ldx, int T_in, const float* w, const float* bias, const float* r, float* y,
                      int64_t ldy, int T_out, int N, int Cin, int taps, int stride, int dil, int pad, int act);

/* MLX affine de-quantisation of a checkpoint tensor at load time (the reference's default checkpoints are 4-bit, group 64:
 * STT/Whisper/WhisperModel.swift:189-196, TTS/Orpheus/TTSEngine/OrpheusWeightLoader.swift:28-60):
 *   out[r][c] = scales[r][c / group] * code[r][c] + biases[r][c / group],  code c of row r = bits [(c % (32/bits)) * bits, ...) of
 *   wq[r][c / (32/bits)] (uint32, little end first).  bits 4 | 8; scales / biases in scale_dtype; out in out_dtype (MIA_F32 | F16 | BF16). */
int mia_dequant_affine(mia_ctx* ctx, const uint32_t* wq, const void* scales, const void* biases, int64_t rows, int64_t cols,
                       int group_size, int bits, int scale_dtype, void* out, int out_dtype, int mem);

/* fp32 scaled-dot-product attention, head dim 64, unmasked, DEVICE pointers (MLXFast.scaledDotProductAttention as called at
 * Codec/S3Gen/Matcha/MatchaTransformer.swift:58-66): q / k / v float32 [B*T][ld*] with head h in columns h*64 .. h*64+63. */
int mia_op_attention_f32(mia_ctx* ctx, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* out,
                         int64_t ldo, int B, int T, int H, float scale);

/* ---- Whisper ---------------------------------------------------------------------------------- */
/* Model dimensions == ModelDimensions (STT/Whisper/Config/WhisperConfig.swift:9-86), read by the caller
 * from config.json. */
typedef struct {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} mia_whisper_dims;

/* A named host tensor handed to the loader (safetensors entry, already mapped by the caller).
 * Names follow the reference's Module key schema (@ModuleInfo keys in STT/Whisper/Layers/ *.swift), e.g.
 *   encoder.conv1.weight [Cout,3,Cin]   encoder.blocks.0.attn.query.weight [D,D]
 *   decoder.token_embedding.weight [V,D]  decoder.positional_embedding [n_text_ctx,D]  decoder.ln.weight [D]
 * `encoder.positional_embedding` is optional (sinusoids, AudioEncoder.swift:78-96, are computed when absent). */
typedef struct {
  const char* name;
  int32_t dtype;      /* mia_dtype of `data` */
  int32_t ndim;
  int64_t shape[4];
  const void* data;   /* host pointer, C-contiguous */
} mia_tensor_view;

/* Decoding options == DecodingOptions + the integer tables the reference's tokenizer provides
 * (STT/Whisper/WhisperDecoding.swift:14-51,104-122,190-206; STT/Whisper/WhisperTokenizer.swift:72-96,489-532).
 * The tokenizer itself (a CPU text codec) stays on the caller's side; its outputs arrive as arrays. */
typedef struct {
  const int32_t* initial_tokens; /* [n_initial] shared by all clips, or [B][n_initial] if per_clip_initial:
                                    ([sot_prev]+prompt)? + sotSequence (+ no_timestamps when timestamps off) */
  int32_t n_initial;
  int32_t per_clip_initial;
  int32_t sot_index;             /* index of <|startoftranscript|> inside initial_tokens (no-speech probe) */
  const int32_t* suppress_ids;   /* nonSpeechTokens + {transcribe, translate, sot, sot_prev, sot_lm, no_speech} */
  int32_t n_suppress;
  const int32_t* blank_ids;      /* tokenizer.encode(" "): suppressed together with eot on the first step only */
  int32_t n_blank;
  int32_t eot, no_speech, no_timestamps, timestamp_begin;
  int32_t timestamps;            /* 0 = TimestampGranularity.none, 1 = segment/word rules active */
  int32_t max_tokens;            /* DecodingOptions.maxTokens (448): budget = max_tokens - n_initial */
  int32_t max_initial_timestamp_index; /* 50 (WhisperDecoding.swift:281) */
  int32_t max_new_tokens;        /* extra cap on generated tokens per clip (0 = none); benchmarking aid */
  float temperature;             /* 0 = greedy (argmax).  > 0: inverse-CDF sampling with caller-provided uniforms */
  const float* uniforms;         /* [B][max_tokens] in [0,1) when any temperature > 0 (explicit RNG), else NULL */
  /* optional per-clip overrides (NULL = use the scalar fields): prefixes of different length (prompt conditioning differs per
   * clip: initial_tokens is then [B][n_initial] with n_initial_per_clip[b] valid entries per row), the fallback temperature
   * of each clip (WhisperSTT.swift:195-250) and which clips take part in this call (inactive clips are left untouched). */
  const int32_t* n_initial_per_clip;
  const int32_t* sot_index_per_clip;
  const float* clip_temperature;
  const int32_t* clip_active;
} mia_decode_opts;

/* Replaces WhisperModel.load's tensor upload (STT/Whisper/WhisperModel.swift:184-206).  compute_dtype is MIA_BF16
 * or MIA_F16 (storage type of weights/activations; accumulation and the residual stream are fp32). */
mia_whisper* mia_whisper_load(mia_ctx* ctx, const mia_whisper_dims* dims, const mia_tensor_view* tensors, int n_tensors,
                              int compute_dtype);
void mia_whisper_free(mia_whisper* w);
/* A second handle on the same (read-only) weights with its own activations, KV caches, decode state and step graph, bound to another
 * context (= another HIP stream) of the same device -- for decoding different batches concurrently (the reference has one model per
 * actor; this is the serving-side counterpart).  Free every clone before the handle it was cloned from. */
mia_whisper* mia_whisper_clone(mia_whisper* src, mia_ctx* ctx);
/* Test hook: force the encoder GEMM tile variant (0 / 1: 128^2 register / LDS-DMA staged, 2: 256^2 two-buffer, 3: automatic
 * (default), 4: 256^2 8-phase).  Results are identical up to fp32 summation order; anything else is MIA_ERR_INVALID_ARGUMENT. */
int mia_whisper_set_gemm_variant(mia_whisper* w, int variant);

/* Test hooks of the decode step (never used by the product path).
 *   mia_whisper_set_debug: bit 0 = launch every step's kernels directly instead of replaying the captured hipGraph, bit 1 = the
 *     one-workgroup decode head at temperature 0 too (normally the split argmax head).  Results must not change.
 *   mia_whisper_trace_logits: from now on every decode step ALSO copies the raw fp32 logits (before the logit rules) of the listed
 *     batch rows into a trace [n_clips][n_text_ctx][n_vocab], filed under the position of the token the step consumed -- the copy is a
 *     node of the same captured step graph, so the trace is what the graph computed.  n_clips = 0 switches it off (<= 8 rows).
 *   mia_whisper_read_logit_trace: rows [first_pos, first_pos + n_pos) of trace slot `slot` -> out (host float32 [n_pos][n_vocab]).
 * Used by tests/ to compare the step path's logits with the oracle's teacher-forced logits at every step (TextDecoder.swift:53-96). */
int mia_whisper_set_debug(mia_whisper* w, int flags);
int mia_whisper_trace_logits(mia_whisper* w, const int32_t* clips, int n_clips);
int mia_whisper_read_logit_trace(mia_whisper* w, int slot, int first_pos, int n_pos, float* out);

/* Replaces model.encode(mel) (WhisperDecoding.swift:98 -> AudioEncoder.swift:43-68) for a batch of 30 s windows and
 * primes the decoder's cross-attention K/V (MultiHeadAttention.swift:49-59).
 *   mel: [B][2*n_audio_ctx][n_mels] in the model's compute dtype. */
int mia_whisper_encode(mia_whisper* w, const void* mel, int B, int mem);
/* Copy the audio features of the last encode ([B][n_audio_ctx][n_audio_state]) out as `dtype` (test hook). */
int mia_whisper_get_audio_features(mia_whisper* w, void* out, int dtype, int mem);

/* Replaces GreedyDecoder.decode's loop (WhisperDecoding.swift:135-359) for the B clips of the last encode:
 * decoder forward with KV cache, logit rules, argmax / sampling, log-prob bookkeeping; all on device.
 *   tokens        int32 [B][max_tokens]: generated tokens (EOT stripped), zero padded
 *   n_tokens      int32 [B]
 *   avg_logprob   float [B]   sum logp / count(non-EOT)            (WhisperDecoding.swift:345-362)
 *   no_speech_prob float [B]  softmax(logits[sot_index])[no_speech] (WhisperDecoding.swift:158-169)
 * Output pointers are host or device according to `mem`. */
int mia_whisper_decode_greedy(mia_whisper* w, const mia_decode_opts* opts, int32_t* tokens, int32_t* n_tokens,
                              float* avg_logprob, float* no_speech_prob, int mem);

/* Language detection (WhisperModel.swift:223-260): one decoder step on [sot]; argmax/softmax over the language
 * token range [sot+1, sot+1+n_languages).  lang_idx int32 [B], prob float [B] (host pointers). */
int mia_whisper_detect_language(mia_whisper* w, int32_t sot, int32_t n_languages, int32_t* lang_idx, float* prob);

/* One 30 s window per clip, end to end (log-mel -> encode -> greedy decode), i.e. the body of the reference's
 * transcribe loop for a batch (WhisperSTT.swift:140-145,181-213).  pcm/offs as in mia_logmel_whisper; pcm and the
 * outputs live in `mem`. */
/* The first half of mia_whisper_transcribe_windows alone: log-mel + encoder (the cross K/V of every decoder layer are primed); follow
 * with mia_whisper_decode_greedy.  Lets a host that drives several handles run their (MFMA-bound) encoders and their
 * (latency-bound) decoders in separate phases: decoders of different batches overlap each other well, an encoder next to a decoder does
 * not. */
int mia_whisper_encode_windows(mia_whisper* w, const float* pcm, const int64_t* offs, int B, int64_t pad_right, int mem);
/* Serving-side scheduling aid: from now on the encoder half of mia_whisper_encode_windows / mia_whisper_transcribe_windows (log-mel,
 * encoder, cross K/V) is enqueued on `hip_stream` instead of the context's stream, ordered against it by events in both directions (the
 * order of a handle's own work does not change).  A host that drives several handles can then create the contexts (decode chains:
 * thousands of short dependent kernels) on HIGH-priority streams and hand every handle a normal-priority encode stream: the decode
 * kernels of one batch are then dispatched ahead of the queued tiles of another batch's encoder instead of taking turns with them.
 * NULL restores the single-stream form.  The stream must belong to the context's device and outlive the handle's use of it. */
int mia_whisper_set_encode_stream(mia_whisper* w, void* hip_stream);
/* Throughput hint for handles that share one weight copy (mia_whisper_clone) and decode CONCURRENTLY on their own streams: with
 * concurrent_readers > 1 the decode step loads its weights with the default cache policy (the loops that read a matrix second and third
 * mostly hit the 256 MB Infinity Cache); with 1 (the default) it streams them non-temporal, which is faster for a lone loop.  Results
 * are bit-identical either way.  Call on every handle of the group; takes effect at the next decode. */
int mia_whisper_set_weight_sharing(mia_whisper* w, int concurrent_readers);
int mia_whisper_transcribe_windows(mia_whisper* w, const float* pcm, const int64_t* offs, int B, int64_t pad_right,
                                   const mia_decode_opts* opts, int32_t* tokens, int32_t* n_tokens, float* avg_logprob,
                                   float* no_speech_prob, int mem);

/* Word-timestamp alignment: the tensor half of findAlignment (STT/Whisper/WhisperTiming.swift:558-748; model.forwardWithCrossQK,
 * STT/Whisper/WhisperModel.swift:95-99).  For every clip of the last encode: a teacher-forced decoder pass over
 * tokens[b][0..n_tokens[b]) = [sot sequence, no_timestamps, text tokens, eot] keeps the pre-softmax cross-attention scores of the
 * alignment heads (heads int32 [n_heads][2] = (layer, head), the model's alignment_heads); on the device: softmax over the first
 * num_frames[b] / 2 frames, standardisation over tokens, width-7 median filter (reflect), mean over heads; on the host like the
 * reference: DTW over rows [row_start, n_tokens[b] - 1) (row_start = index of no_timestamps) of the negated matrix.
 *   token_probs [B][stride]: P(tokens[p + 1]) under softmax(logits[p][0:eot]);  text_idx / time_idx [B][path_cap] + path_len [B]: the
 *   DTW path (row index relative to row_start, frame index);  matrix (nullable) [B][stride][n_audio_ctx]: the filtered, head-averaged
 *   weights.  All host pointers.  Word splitting and the jump-time arithmetic (:735-820) stay with the caller (tokenizer text). */
int mia_whisper_align(mia_whisper* w, const int32_t* tokens, int stride, const int32_t* n_tokens, const int32_t* heads, int n_heads,
                      const int32_t* num_frames, int row_start, int eot, float* token_probs, int32_t* text_idx, int32_t* time_idx,
                      int32_t* path_len, int path_cap, float* matrix);

/* ---- neural codec decoders (fp32, like the reference) ------------------------------------------ */
/* SNACConfig (TTS/Orpheus/SNAC/SNACConfig.swift:9-94); tensors use the checkpoint key schema the reference loads
 * (decoder.model.layers.N..., quantizer.quantizers.i.{codebook.weight,out_proj.{weight_g,weight_v,bias}};
 * SNACDecoder.swift:101-243,358-368), float32. */
typedef struct {
  int32_t latent_dim, decoder_dim;
  int32_t n_rates; int32_t decoder_rates[8];
  int32_t n_vq; int32_t vq_strides[4];
  int32_t codebook_size, codebook_dim;
  int32_t noise, depthwise;
} mia_snac_config;
/* DACConfig (Codec/DAC/DACModel.swift:169-203): decoder side only. */
typedef struct {
  int32_t latent_dim, decoder_dim;
  int32_t n_rates; int32_t decoder_rates[8];
  int32_t n_codebooks, codebook_size, codebook_dim;
} mia_dac_config;

mia_codec* mia_snac_load(mia_ctx* ctx, const mia_snac_config* cfg, const mia_tensor_view* tensors, int n_tensors);
mia_codec* mia_dac_load(mia_ctx* ctx, const mia_dac_config* cfg, const mia_tensor_view* tensors, int n_tensors);
void mia_codec_free(mia_codec* c);
/* Samples produced / Gaussian values consumed for a latent of `latent_len` steps (SNAC: max_i n_i*vq_stride_i; DAC: T). */
int64_t mia_codec_output_len(mia_codec* c, int64_t latent_len);
int64_t mia_codec_noise_len(mia_codec* c, int64_t latent_len);
/* Replaces SNACDecoder.decode(codes:) (TTS/Orpheus/SNAC/SNACDecoder.swift:281-289, called at
 * TTS/Orpheus/TTSEngine/OrpheusTTS.swift:361).  codes[i] -> n_codes[i] ids of VQ level i.  `noise`: the N(0,1) draws of
 * the NoiseBlocks (NoiseBlock.swift:33), block after block, mia_codec_noise_len() values in total, or NULL for none.
 * pcm: float32 mono 24 kHz. */
int mia_snac_decode(mia_codec* c, const int32_t* const* codes, const int32_t* n_codes, int n_levels, const float* noise,
                    int64_t n_noise, float* pcm, int64_t pcm_capacity, int64_t* n_samples, int mem);
/* Replaces DACCodec.decodeFromCodes (Codec/DAC/DACModel.swift:303-306) for one sequence: codes int32 [n_codebooks][T]. */
int mia_dac_decode(mia_codec* c, const int32_t* codes, int n_codebooks, int64_t T, float* pcm, int64_t pcm_capacity,
                   int64_t* n_samples, int mem);
/* Encoder side of the DAC codec: replaces DACCodec.encode (Codec/DAC/DACModel.swift:284-296): preprocess (right-pad to the hop
 * length :308-317) -> DACEncoder (:43-86; blocks :15-38) -> residual vector quantisation (DACQuantize.swift:147-190, nearest entry
 * on L2-normalised vectors :87-115, first index on ties).  mia_dac_load_encoder attaches the `encoder.*` and
 * `quantizer.quantizers.N.in_proj.*` tensors to a handle made by mia_dac_load (encoder_dim * 2^n_rates must equal its latent width).
 * mia_dac_encode: pcm float32 mono [n_samples] -> codes int32 [n_quantizers][codes_capacity] (row q holds *n_steps ids);
 * n_quantizers <= 0 = all codebooks.  mia_dac_code_len(n_samples) = steps produced. */
typedef struct {
  int32_t encoder_dim;
  int32_t n_rates; int32_t encoder_rates[8];
} mia_dac_encoder_config;
int mia_dac_load_encoder(mia_codec* c, const mia_dac_encoder_config* cfg, const mia_tensor_view* tensors, int n_tensors);
int64_t mia_dac_code_len(mia_codec* c, int64_t n_samples);
int mia_dac_encode(mia_codec* c, const float* pcm, int64_t n_samples, int n_quantizers, int32_t* codes, int64_t codes_capacity,
                   int64_t* n_steps, int mem);

/* ---- autoregressive LMs (Llama-3 / Qwen2 blocks) ------------------------------------------------ */
/* OrpheusConfig (TTS/Orpheus/BuildingBlocks/TransformerBlock.swift:16-34) / Qwen2Config (TTS/CosyVoice2/LLM/Qwen2LM.swift:15-43).
 * Tensors use the Hugging Face key schema both ports load: model.embed_tokens.weight, model.layers.N.self_attn.{q,k,v,o}_proj.weight
 * (+ .bias for q,k,v when qkv_bias), model.layers.N.mlp.{gate,up,down}_proj.weight, model.layers.N.{input,post_attention}_layernorm.weight,
 * model.norm.weight, lm_head.weight (absent when tie_embeddings). */
typedef struct {
  int32_t vocab, hidden, inter, n_layers, n_heads, n_kv_heads, head_dim, max_ctx;
  float rms_eps, rope_theta;
  int32_t rope_llama3;            /* 1: Llama3RoPE frequency scaling (TTS/Shared/Llama3RoPE.swift:27-66) */
  float rope_factor, rope_low, rope_high;
  int32_t rope_old_ctx;
  int32_t qkv_bias, tie_embeddings;
} mia_lm_config;

/* sampleNextToken parameters (TTS/Orpheus/TTSEngine/OrpheusTTS.swift:375-470): repetition penalty over the last
 * rep_window generated ids -> /temperature -> top-p (keeps the first token crossing p) -> categorical. */
typedef struct {
  float temperature, top_p, rep_penalty;
  int32_t rep_window, max_new_tokens;
  int32_t n_stop; int32_t stop_ids[4];
  int32_t reserved;
} mia_lm_sampler;

mia_lm* mia_lm_load(mia_ctx* ctx, const mia_lm_config* cfg, const mia_tensor_view* tensors, int n_tensors, int dtype);
/* MLX-affine quantised (group 64; 4- or 8-bit codes) weights for the decode step: the reference's default Orpheus / CosyVoice2
 * checkpoints are quantised (TTS/Orpheus/TTSEngine/OrpheusWeightLoader.swift:28-60; the bit width is a loader option,
 * Models/TranscriptionResult.swift:162-198) and MLX multiplies them packed (quantizedMatmul).  mia_lm_attach_quantized takes every step
 * Linear as stored -- `<name>.weight` (MIA_U32 codes [N][K * bits / 32], little end first), `<name>.scales`, `<name>.biases` ([N][K/64],
 * both f16 or both bf16) -- onto a handle loaded from the de-quantised checkpoint; the per-token step then streams the packed codes and
 * multiplies them as MLX's own kernels do (per group: scale * sum(code * x) + bias * sum(x), the codes exact in the MFMA's 16-bit
 * operands, fp32 group sums), the batched prompt pass keeps the 16-bit copy.  One attach per handle.  mia_lm_attach_q4 = bits 4.
 * mia_lm_use_q4 toggles the step between the packed and the 16-bit weights (0 = 16-bit). */
int mia_lm_attach_quantized(mia_lm* lm, const mia_tensor_view* tensors, int n_tensors, int group_size, int bits);
int mia_lm_attach_q4(mia_lm* lm, const mia_tensor_view* tensors, int n_tensors, int group_size);
int mia_lm_use_q4(mia_lm* lm, int on);
/* Test hook: bit 0 = launch every step's kernels directly (no hipGraph), bit 1 = feed prompts token by token (no batched prompt
 * pass).  Results agree up to fp32 summation order. */
int mia_lm_set_debug(mia_lm* lm, int flags);
void mia_lm_free(mia_lm* lm);
int mia_lm_reset(mia_lm* lm);
/* model(ids, cache) then logits[0,-1] (OrpheusTTS.swift:245-251,289): appends n ids to the KV cache, returns the fp32
 * logits after the last one into last_logits [vocab] (host pointer, may be NULL). */
int mia_lm_forward(mia_lm* lm, const int32_t* ids, int n, float* last_logits);
/* The whole sampling loop of generateChunk (OrpheusTTS.swift:245-348) on device: prompt, then up to max_new_tokens sampled
 * ids (stops after emitting a stop id).  uniforms: max_new_tokens values in [0,1), one per draw (explicit RNG: inverse CDF
 * over the kept tokens in index order).  Host pointers. */
int mia_lm_generate(mia_lm* lm, const int32_t* prompt, int n_prompt, const mia_lm_sampler* sampler, const float* uniforms,
                    int32_t* out_tokens, int32_t* n_out);
/* Sentence-level batching (MI355X-side addition; the reference generates its sentences one after another, OrpheusTTS.swift:179-191):
 * mia_lm_set_batch sizes the per-sequence state (K/V caches, repetition windows, ...) for up to max_batch <= 32 sequences;
 * mia_lm_generate_batch runs mia_lm_generate's loop for n_seq prompts side by side -- one read of the weights per step for all of
 * them.  prompts: all ids back to back, prompt_offsets [n_seq + 1]; uniforms [n_seq][max_new_tokens]; out_tokens
 * [n_seq][max_new_tokens]; n_out [n_seq].  Sequence b's output equals mia_lm_generate(prompt b, uniforms row b) on the same handle
 * capacity: a sequence's ids never depend on the batch it sits in.  (The capacity selects the step's kernel chain -- up to 4 sequences
 * the low-latency chain with the RMSNorm carried across the GEMMs, above that the split-K chain -- so runs under capacities on
 * different sides of 4 agree to fp32 summation order, not bit for bit.)  Host pointers. */
int mia_lm_set_batch(mia_lm* lm, int max_batch);
int mia_lm_generate_batch(mia_lm* lm, const int32_t* prompts, const int32_t* prompt_offsets, int n_seq, const mia_lm_sampler* sampler,
                          const float* uniforms, int32_t* out_tokens, int32_t* n_out);
/* CosyVoice2 RAS sampling parameters (TTS/CosyVoice2/LLM/Qwen2LM.swift:433-488 defaults: top_p 0.8, top_k 25, win 10, tau 0.1;
 * eos = speech_token_size (6561); min_len / max_len = 2x / 20x the text length, :368-372). */
typedef struct { float top_p; int32_t top_k; int32_t win; float tau; int32_t eos; int32_t min_len; int32_t max_len; } mia_ras_params;
/* Qwen2LM.inference + inferenceLoop (Qwen2LM.swift:335-427) on device.  The model must have been loaded with the extra
 * tensors llm_decoder.{weight,bias} and speech_embedding.weight.  prompt_embeds: float32 [n_prompt][hidden] =
 * [llm_embedding[sos], embed_tokens(prompt_text + text), llm_embedding[task], speech_embedding(prompt tokens)], gathered by
 * the caller.  uniforms: the stream of explicit draws (one per categorical; rejected EOS trials consume more).  Host pointers. */
int mia_lm_generate_ras(mia_lm* lm, const float* prompt_embeds, int n_prompt, const mia_ras_params* rp, const float* uniforms,
                        int n_uniforms, int32_t* out_tokens, int32_t* n_out);
/* mia_lm_generate_ras for n_seq utterances side by side (after mia_lm_set_batch): prompt_embeds = the rows of all prompts back to back
 * [prompt_offsets[n_seq]][hidden]; rp [n_seq] (top_p, top_k, win, tau, eos must agree; min_len / max_len are per utterance); uniforms
 * [n_seq][n_uniforms]; out_tokens [n_seq][out_stride], out_stride >= max(max_len) + 1; n_out [n_seq].  Utterance b's ids equal
 * mia_lm_generate_ras(prompt b, rp[b], uniforms row b).  Host pointers. */
int mia_lm_generate_ras_batch(mia_lm* lm, const float* prompt_embeds, const int32_t* prompt_offsets, int n_seq, const mia_ras_params* rp,
                              const float* uniforms, int n_uniforms, int32_t* out_tokens, int out_stride, int32_t* n_out);
/* One sampleNextToken call on caller-provided logits (host pointers). */
int mia_sample_top_p(mia_ctx* ctx, const float* logits, int V, const int32_t* history, int n_hist, float rep_penalty,
                     float temperature, float top_p, float uniform, int32_t* out);

/* ---- S3Tokenizer (speech -> 25 Hz token ids) ---------------------------------------------------- */
/* S3TokenizerModelConfig / V3 (Codec/S3Tokenizer/S3TokenizerConfig.swift:9-90): V2 = 6 blocks, V3 = 12. */
typedef struct { int32_t n_mels, n_audio_state, n_audio_head, n_audio_layer; } mia_s3_config;
/* float32 tensors, Module key schema of the reference: encoder.conv{1,2}.{weight,bias}, encoder.blocks.N.attn.{query,key,value,out}.*,
 * encoder.blocks.N.attn.fsmn_block.weight, encoder.blocks.N.{attn_ln,mlp_ln}.*, encoder.blocks.N.mlp.layers.{0,2}.*,
 * quantizer.fsq_codebook.project_down.{weight,bias}. */
mia_s3tok* mia_s3tok_load(mia_ctx* ctx, const mia_s3_config* cfg, const mia_tensor_view* tensors, int n_tensors);
void mia_s3tok_free(mia_s3tok* s3);
/* Replaces S3TokenizerV2.quantize(mel:melLen:) for clips of at most 30 s (Codec/S3Tokenizer/S3Tokenizer.swift:474-494; called at
 * TTS/CosyVoice2/CosyVoice2TTS.swift:391).  mel float32 [B][n_mels][T] (the layout mia_logmel_s3 emits), mel_len / tok_len host
 * int32 [B]; tokens int32 [B][tokens_stride], zero padded; token count = ((len-1)/2+1 - 1)/2 + 1. */
int mia_s3tok_encode(mia_s3tok* s3, const float* mel, const int32_t* mel_len, int B, int T, int32_t* tokens, int tokens_stride,
                     int32_t* tok_len, int mem);

/* ---- HiFT vocoder (80-bin mel @ 50 Hz -> 24 kHz waveform) ------------------------------------------ */
/* Constructor arguments of CosyHiFTGenerator (TTS/CosyVoice2/HiFiGAN/CosyHiFTGenerator.swift:283-300).  n_fft 16 / hop 4 are
 * fixed (istftParams).  The residual-block dilations are shared by every block ([1,3,5] in the reference). */
typedef struct {
  int32_t in_channels, base_channels, nb_harmonics, sampling_rate;
  int32_t n_ups; int32_t up_rates[4]; int32_t up_kernels[4];
  int32_t n_res_kernels; int32_t res_kernels[4];
  int32_t src_res_kernels[4];          /* one per upsample stage */
  int32_t n_dilations; int32_t dilations[4];
  float nsf_alpha, nsf_sigma, voiced_threshold, lrelu_slope, audio_limit;
} mia_hift_config;
typedef struct mia_hift mia_hift;
/* float32 tensors, Module key schema of the reference (CosyHiFTGenerator.swift:272-279): f0_predictor.condnet_{0,2,4,6,8}.*,
 * f0_predictor.classifier.*, m_source.l_linear.*, conv_pre.*, ups.N.*, source_downs.N.*, source_resblocks.N.{convs1,convs2}.M.*,
 * source_resblocks.N.{activations1,activations2}.M.alpha, resblocks.N.(same), conv_post.*  (Conv1d weight [Cout][K][Cin],
 * ConvTransposed1d weight [Cout][K][Cin]). */
mia_hift* mia_hift_load(mia_ctx* ctx, const mia_hift_config* cfg, const mia_tensor_view* tensors, int n_tensors);
void mia_hift_free(mia_hift* h);
/* samples produced per mel frame: prod(up_rates) * hop (480) */
int mia_hift_upsample_factor(const mia_hift* h);
/* CosyF0Predictor (CosyHiFTGenerator.swift:236-255): mel float32 [in_channels][T] (channel-major as in the reference) -> f0 [T]. */
int mia_hift_f0(mia_hift* h, const float* mel, int T, float* f0, int mem);
/* f0Upsample + SourceModuleHnNSF2 / SineGen2 (CosyHiFTGenerator.swift:96-153,190-203,408-410,487-495): f0 [T] -> source [480 T].
 * noise: the reference's MLXRandom.normal draw, float32 [480 T][nb_harmonics+1], or NULL for no additive noise.  The reference's
 * random initial phase (:103-111) only touches sample 0 of the full-rate `rad` track, which the 480:1 linear downsample never
 * reads (it reads samples 480 i + 239 / + 240), so it has no effect on the output and is not an input here.  T <= 17000. */
int mia_hift_source(mia_hift* h, const float* f0, int T, const float* noise, float* source, int mem);
/* CosyHiFTGenerator.decode(x:s:) (:412-475): mel [in_channels][T] + source [480 T] -> waveform [480 T], clipped to +-audio_limit. */
int mia_hift_decode(mia_hift* h, const float* mel, int T, const float* source, float* pcm, int mem);
/* CosyHiFTGenerator.callAsFunction (:482-505): f0 -> source (its first cache_len samples replaced by cache_source, the
 * streaming hand-over) -> decode.  source_out (nullable) receives the source actually used.  Every buffer lives in `mem`;
 * with MIA_MEM_DEVICE nothing is copied and the call returns after enqueueing on the ctx stream. */
int mia_hift_vocode(mia_hift* h, const float* mel, int T, const float* noise, const float* cache_source, int cache_len,
                    float* pcm, float* source_out, int mem);
/* mia_hift_vocode for n utterances in one pass (an MI355X-side addition, like mia_flow_inference_batch: the reference vocodes its
 * utterances one after another, CosyVoice2Model.swift:200-208).  mels: the utterances' [in_channels][T[u]] blocks back to back; noise
 * (nullable): their [480 T[u]][nb_harmonics+1] blocks back to back; pcm: 480 T[u] samples per utterance back to back.  Every
 * convolution runs once over the stacked sequences (a sequence's rows past its own end read as zero, the padding its single call
 * sees); each utterance's waveform equals its own mia_hift_vocode call bit for bit.  1 <= n <= 64.  The cache_source hand-over of
 * the streaming path stays with the single-utterance call.  Returns after the waveforms are in `pcm`. */
int mia_hift_vocode_batch(mia_hift* h, const float* mels, const int32_t* T, int n, const float* noise, float* pcm, int mem);

/* ---- CosyVoice2 flow (speech tokens -> 80-bin mel @ 50 Hz) ---------------------------------------- */
/* FlowConfig (TTS/CosyVoice2/Config/CosyVoice2Config.swift:79-127).  Built for the shapes the reference engine uses: head dim 64 in
 * both the conformer encoder (input_size = 64 * enc_heads) and the estimator, mel 80, dec_in_channels = 4 * 80, one U-Net level. */
typedef struct {
  int32_t input_size, output_size, spk_embed_dim, vocab_size, pre_lookahead_len, n_timesteps;
  int32_t enc_heads, enc_linear_units, enc_blocks, enc_up_blocks, upsample_stride;
  int32_t dec_in_channels, dec_channels, dec_heads, dec_n_blocks, dec_mid_blocks;
  float cfg_rate;   /* inference_cfg_rate, 0.7 */
} mia_flow_config;
typedef struct mia_flow mia_flow;
/* float32 tensors under the reference's (remapped) Module keys (CosyVoice2TTS.swift:320-336): input_embedding.weight,
 * spk_embed_affine_layer.*, encoder.{embed,up_embed}.{linear,norm}.*, encoder.pre_lookahead_layer.conv{1,2}.*, encoder.up_layer.conv.*,
 * encoder.{encoders,up_encoders}.N.{self_attn.{linear_q,linear_k,linear_v,linear_out,linear_pos}.*, self_attn.pos_bias_{u,v},
 * feed_forward.{w_1,w_2}.*, norm_mha.*, norm_ff.*}, encoder.after_norm.*, encoder_proj.*, decoder.estimator.time_mlp.linear_{1,2}.*,
 * decoder.estimator.{down_blocks.0,mid_blocks.N,up_blocks.0}.{resnet.{mlp_linear,block{1,2}.conv.conv,block{1,2}.norm,res_conv}.*,
 * transformers.K.{norm1,norm3,attn.{query_proj,key_proj,value_proj,out_proj},ff.layers.{0,1}}.*}, ...down_blocks.0.downsample.conv.*,
 * ...up_blocks.0.upsample.conv.*, decoder.estimator.final_block.{conv.conv,norm}.*, decoder.estimator.final_proj.*. */
mia_flow* mia_flow_load(mia_ctx* ctx, const mia_flow_config* cfg, const mia_tensor_view* tensors, int n_tensors);
void mia_flow_free(mia_flow* f);
/* input_embedding -> UpsampleConformerEncoder(streaming: false) -> encoder_proj (CosyVoice2Model.swift:496-522,
 * UpsampleConformerEncoder.swift:407-474): token int32 [n] -> mu float32 [upsample_stride * n][80]. */
int mia_flow_encode(mia_flow* f, const int32_t* token, int n_token, float* mu, int mem);
/* CosyVoice2FlowModule.inference(finalize: true) (CosyVoice2Model.swift:467-553) for one utterance:
 *   token [n_token], prompt_token [n_prompt] int32; prompt_feat float32 [prompt_feat_len][80] (the prompt's mel, time-major as in the
 *   reference); embedding float32 [spk_embed_dim]; z float32 [80][T], T = upsample_stride (n_token + n_prompt): the CFM's initial
 *   noise, which the reference draws internally (CosyVoice2CFM.swift:86) -- explicit here; n_timesteps <= 0 means cfg.n_timesteps.
 *   mel out float32 [80][T - prompt_feat_len] (channel-major, what CosyHiFTGenerator consumes).  Every buffer lives in `mem`. */
int mia_flow_inference(mia_flow* f, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt, const float* prompt_feat,
                       int prompt_feat_len, const float* embedding, const float* z, int n_timesteps, float* mel, int mem);
/* n_utt utterances through ONE pass of the flow (an MI355X-side addition like mia_lm_generate_ras_batch: the reference synthesises its
 * sentences one after another, CosyVoice2Model.swift:155-208,467-553).  Every argument of mia_flow_inference becomes an array indexed by
 * utterance (prompt_token / prompt_feat may be null when every utterance has none; entry u may be null when n_prompt[u] /
 * prompt_feat_len[u] is 0); mel[u] receives float32 [80][T_u - prompt_feat_len[u]], T_u = upsample_stride (n_token[u] + n_prompt[u]).
 * The utterances run as stacked sequences padded to the longest one; padding never reaches a valid frame (per-sequence key masks and
 * zero look-ahead), so mel[u] is bit-identical to the utterance's own mia_flow_inference call.  1 <= n_utt <= 64. */
int mia_flow_inference_batch(mia_flow* f, int n_utt, const int32_t* const* token, const int32_t* n_token, const int32_t* const* prompt_token,
                             const int32_t* n_prompt, const float* const* prompt_feat, const int32_t* prompt_feat_len, const float* const* embedding,
                             const float* const* z, int n_timesteps, float* const* mel, int mem);
/* The same call with the two switches the modules carry for chunked synthesis:
 *   finalize = 0 drops the encoder's last pre_lookahead_len * upsample_stride frames before the CFM (CosyVoice2Model.swift:504-510), so
 *     T = upsample_stride (n_token + n_prompt) - that trim; z is [80][T] for THAT T and *mel_frames = T - prompt_feat_len;
 *   enc_static_chunk / dec_static_chunk > 0 turn on the block-causal "streaming" attention masks (query i sees keys below
 *     (i / chunk + 1) chunk: subsequentChunkMask, Codec/S3Gen/Transformer/UpsampleConformerEncoder.swift:124-195) of the conformer
 *     encoder (chunk, and chunk * upsample_stride after the up-sampling, :424-460) and of the estimator's transformer blocks
 *     (Codec/S3Gen/S3GenDecoder.swift:304-320); the checkpoint's values are encoder_static_chunk_size / decoder_static_chunk_size
 *     (TTS/CosyVoice2/Config/CosyVoice2Config.swift:155,166).  0 / 0 with finalize = 1 is mia_flow_inference. */
int mia_flow_inference_streaming(mia_flow* f, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt, const float* prompt_feat,
                                 int prompt_feat_len, const float* embedding, const float* z, int n_timesteps, int finalize, int enc_static_chunk,
                                 int dec_static_chunk, float* mel, int* mel_frames, int mem);

/* ---- CAM++ speaker encoder (CosyVoice2 prepareConditionals, once per speaker) ---------------------------------
 * Replaces CAMPlusSpeakerEncoder (TTS/CosyVoice2/SpeakerEncoder/CAMPlusSpeakerEncoder.swift:12-150, called at
 * TTS/CosyVoice2/CosyVoice2TTS.swift:409) = Codec/S3Gen/CAMPPlus.swift: kaldiFbankCAMPPlus :32-108, CAMPPlus :687-785,
 * CAMPPlus.inference :788-818, in the one configuration that wrapper builds (80 fbank bins, 192-d embedding, growth 32,
 * bottleneck 4 x 32, 128 initial channels, "batchnorm-relu", segment output).  BatchNorm uses its running statistics. */
#define MIA_CAMPPLUS_DIM 192
typedef struct mia_campplus mia_campplus;
/* float32 tensors with the reference's Module key paths ("campplus." prefix already stripped, CAMPlusSpeakerEncoder.swift:93-104):
 * head.{conv1,conv2}.weight [32][3][3][Cin], head.{bn1,bn2}.{weight,bias,running_mean,running_var},
 * head.layer{1,2}.{0,1}.{conv1,conv2}.weight, .{bn1,bn2}.*, head.layer{1,2}.0.shortcut.{0.weight [32][1][1][32], 1.*},
 * tdnn.linear.weight [128][5][320], tdnn.nonlinear.0.*, blocks.B.layers.I.{nonlinear1.0.*, linear1.weight [128][1][Cin],
 * nonlinear2.0.*, cam_layer.linear_local.weight [32][3][128], cam_layer.linear1.{weight [64][1][128], bias},
 * cam_layer.linear2.{weight [32][1][64], bias}}, transits.B.{nonlinear.0.*, linear.weight}, out_nonlinear.0.*,
 * dense.linear.weight [192][1][1024], dense.nonlinear.0.{running_mean,running_var}. */
mia_campplus* mia_campplus_load(mia_ctx* ctx, const mia_tensor_view* tensors, int n_tensors);
void mia_campplus_free(mia_campplus* m);
/* kaldiFbankCAMPPlus (CAMPPlus.swift:32-108): pcm float32 mono 16 kHz [n_samples >= 400] -> fbank float32
 * [mia_kaldi_fbank_frames(n_samples)][80] (frame-major).  mean_norm != 0 also removes each bin's mean over time (:797-799),
 * which is what the encoder consumes; 0 gives extractFbank's raw features (CAMPlusSpeakerEncoder.swift:138-150). */
int64_t mia_kaldi_fbank_frames(int64_t n_samples);
int mia_campplus_fbank(mia_campplus* m, const float* pcm, int64_t n_samples, int mean_norm, float* fbank, int mem);
/* CAMPPlus.callAsFunction (:755-785) for one clip: feats float32 [n_frames][80] -> emb float32 [192]. */
int mia_campplus_forward(mia_campplus* m, const float* feats, int n_frames, float* emb, int mem);
/* CAMPPlus.inference (:788-818) for one clip: pcm float32 mono 16 kHz [n_samples] -> emb float32 [192]. */
int mia_campplus_embed(mia_campplus* m, const float* pcm, int64_t n_samples, float* emb, int mem);

#ifdef __cplusplus
}
#endif
#endif /* MIA_H */
