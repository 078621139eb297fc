"""Codec-only timing (SNAC / DAC / HiFT decode) -- the `codec` object of bench.py without the Whisper pass; used under rocprofv3."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mlx_swift_audio_amd as M  # noqa: E402

if __name__ == "__main__":
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    ctx = M.Context(stream=st.cuda_stream)
    print(json.dumps(bench.codec_bench(ctx, torch)))
