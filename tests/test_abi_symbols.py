"""The C-ABI library loads on a machine without a GPU and exports every symbol include/mia.h declares; compute entry
points fail loudly (MIA_ERR_DEVICE) instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mia.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mia_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    import mlx_swift_audio_amd as m
    lib = m._lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mia.h but not exported by libmia.so"
    assert lib.mia_version().startswith(b"mia ")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import mlx_swift_audio_amd as m
    with pytest.raises(m.MiaError) as e:
        m.Context(0)
    assert e.value.code == m._lib.ERR_DEVICE
    lib = m._lib.load()
    assert lib.mia_create(0) in (None, 0)
    lib.mia_last_error.restype = ctypes.c_char_p
    assert lib.mia_last_error(None) == b"null ctx"
