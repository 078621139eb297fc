"""In-tree build of the gfx950 HIP layer: csrc/*.hip -> lib/libmia.so (explicit hipcc, no JIT cache).

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmia.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), _headers_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 4) -> str:
    """Compile every HIP source for gfx950 and link lib/libmia.so. Returns the library path."""
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
