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

typedef enum {
  MIA_OK = 0,
  MIA_ERR_INVALID_ARGUMENT = -1, /* STTError.invalidArgument */
  MIA_ERR_MODEL_NOT_LOADED = -2, /* STTError.modelNotLoaded */
  MIA_ERR_INVALID_AUDIO = -3,    /* STTError.invalidAudio */
  MIA_ERR_OUT_OF_MEMORY = -4,
  MIA_ERR_DEVICE = -5,           /* no usable gfx950 device / HIP failure */
  MIA_ERR_UNSUPPORTED = -6
} mia_status;

typedef enum { MIA_F32 = 0, MIA_F16 = 1, MIA_BF16 = 2 } mia_dtype;
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

#ifdef __cplusplus
}
#endif
#endif /* MIA_H */
