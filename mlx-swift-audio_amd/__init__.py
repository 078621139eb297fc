"""mlx-swift-audio_amd -- MI355X-native speech-inference hot path (log-mel, Whisper encode/decode, codecs).

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), lib/ (built libmia.so) and the
host-side mirror of the reference interface.  Import as `mlx_swift_audio_amd` (see the shim at repo root).
"""
from . import _lib
from ._lib import BF16, F16, F32, Context, MiaError

__all__ = ["_lib", "Context", "MiaError", "F32", "F16", "BF16"]
