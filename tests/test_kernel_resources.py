"""Occupancy guards (no GPU needed: hipcc cross-compiles gfx950): kernels whose speed depends on fitting 3 waves per SIMD must stay at or
under 168 VGPRs -- 512 / 168 = 3 waves, 172 already means 2 (round 3: four VGPRs of per-element row arithmetic in the tap GEMM's
epilogue cost SNAC 10 % and DAC 15 % until the measurement showed it)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _vgprs(src):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "mlx-swift-audio_amd", "csrc", src), "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        text = open(out).read()
    res = {}
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", text, re.S):
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        res[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)), int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)))
    return res


def test_tap_gemm_128x128_tile_keeps_three_waves_per_simd():
    res = _vgprs("codec_kernels.hip")
    big = [v for k, v in res.items() if "conv_gemm_f32ILi2ELi2ELi1E" in k]
    assert big, sorted(res)
    for vgpr, spill in big:
        assert vgpr <= 168 and spill == 0, (vgpr, spill)


def test_encoder_attention_keeps_three_waves_per_simd():
    res = _vgprs("attention.hip")
    ks = [v for k, v in res.items() if "enc_attention_kernel" in k]
    assert ks, sorted(res)
    for vgpr, spill in ks:
        assert vgpr <= 168 and spill == 0, (vgpr, spill)
