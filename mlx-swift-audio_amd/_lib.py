"""ctypes binding of lib/libmia.so (the C ABI in include/mia.h).

The product path never falls back to a CPU implementation: if the library is missing this module raises
at load time, and if no gfx950 device is present `Context()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmia.so")

MIA_OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_MODEL_NOT_LOADED = -2
ERR_INVALID_AUDIO = -3
ERR_OUT_OF_MEMORY = -4
ERR_DEVICE = -5
ERR_UNSUPPORTED = -6

F32, F16, BF16, U32 = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1


class MiaError(RuntimeError):
    """Mirrors the reference's STTError/TTSError cases (Models/STTError.swift:6-46) by status code."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"mia error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libmia.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    _declare(lib)
    _lib = lib
    return lib


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.mia_version.restype = C.c_char_p
    lib.mia_version.argtypes = []
    lib.mia_create.restype = vp
    lib.mia_create.argtypes = [i32]
    lib.mia_create_on_stream.restype = vp
    lib.mia_create_on_stream.argtypes = [i32, vp]
    lib.mia_destroy.restype = None
    lib.mia_destroy.argtypes = [vp]
    lib.mia_last_error.restype = C.c_char_p
    lib.mia_last_error.argtypes = [vp]
    lib.mia_stream.restype = vp
    lib.mia_stream.argtypes = [vp]
    lib.mia_synchronize.restype = i32
    lib.mia_synchronize.argtypes = [vp]
    lib.mia_profile_enable.restype = i32
    lib.mia_profile_enable.argtypes = [vp, i32]
    lib.mia_profile_reset.restype = i32
    lib.mia_profile_reset.argtypes = [vp]
    lib.mia_profile_read.restype = i32
    lib.mia_profile_read.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for name in ("mia_logmel_whisper", "mia_logmel_s3"):
        f = getattr(lib, name)
        f.restype = i32
        f.argtypes = [vp, vp, vp, i32, i32, i64, i64, vp, i32, i32]


class Context:
    """One device + one HIP stream; calls are serialised by the caller (the reference's actor model)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load()
        if stream is None:
            self.h = self.lib.mia_create(device)
        else:
            self.h = self.lib.mia_create_on_stream(device, C.c_void_p(stream))
        if not self.h:
            raise MiaError(ERR_DEVICE, f"mia_create({device}) failed: no usable gfx950 device (there is no CPU fallback)")
        self.device = device

    def check(self, rc: int) -> None:
        if rc != MIA_OK:
            raise MiaError(rc, self.lib.mia_last_error(self.h).decode())

    def profile(self, on: bool) -> None:
        self.check(self.lib.mia_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self) -> None:
        self.check(self.lib.mia_profile_reset(self.h))

    def profile_read(self, kernel_class: str) -> tuple[int, float, float]:
        """(launches, total milliseconds, total work) of a kernel class since the last reset."""
        n, ms, work = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        self.check(self.lib.mia_profile_read(self.h, kernel_class.encode(), C.byref(n), C.byref(ms), C.byref(work)))
        return n.value, ms.value, work.value

    def synchronize(self) -> None:
        self.check(self.lib.mia_synchronize(self.h))

    @property
    def stream(self) -> int:
        return int(self.lib.mia_stream(self.h) or 0)

    def adopt(self, child) -> None:
        """Model handles register here: they must be released BEFORE the context (their free() synchronises on its stream).  Weak
        references: adoption never extends a model's life."""
        import weakref
        if not hasattr(self, "_children"):
            self._children = []
        self._children.append(weakref.ref(child))

    def close(self) -> None:
        if getattr(self, "h", None):
            for ref in getattr(self, "_children", []):
                obj = ref()
                if obj is not None:
                    try:
                        obj.close()
                    except Exception:
                        pass
            self._children = []
            self.lib.mia_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
