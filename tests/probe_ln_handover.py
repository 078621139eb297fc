"""(Run by hand on the GPU box: `python tests/probe_ln_handover.py`; lives under tests/ because it uses the oracle.)
Precision probe of the encoder's LayerNorm hand-over (gemm.h): the 16-bit x * gamma operand carries the row's common-mode value, so
its rounding error grows with |row mean| / row std.  Shifts the residual stream's mean through conv2's bias and prints the encoder
feature error against the fp32 oracle with the hand-over (auto variant) and without it (forced variant 2: LayerNorm kernel)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import whisper as HW
from oracle import whisper as OW

ctx = m.Context()
dims = OW.DIMS["micro"]
for dtype_name, dt in (("bf16", m.BF16), ("f16", m.F16)):
    for off in (0.0, 0.5, 2.0, 8.0):
        w = OW.synthetic_weights(dims, seed=5, round_to=None)
        w = dict(w)
        w["encoder.conv2.bias"] = (np.asarray(w["encoder.conv2.bias"], np.float32) + off).astype(np.float32)
        w = {k: OW.round_array(np.asarray(v, np.float32), dtype_name) if np.asarray(v).dtype.kind == "f" else v for k, v in w.items()}
        ora = OW.WhisperOracle(dims, w)
        rng = np.random.default_rng(0)
        mel = OW.round_array((0.5 * rng.standard_normal((2, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32), dtype_name)
        ref = ora.encode(mel).numpy()
        out = []
        for variant in (3, 2):
            model = HW.WhisperModel.load(ctx, dims, w, dt)
            model.set_gemm_variant(variant)
            model.encode(mel)
            got = model.audio_features()
            err = np.abs(got - ref)
            out.append((err.max(), err.mean()))
            model.close()
        print(f"{dtype_name} conv2 bias +{off}: hand-over max {out[0][0]:.4f} mean {out[0][1]:.5f} | LayerNorm kernel max {out[1][0]:.4f} mean {out[1][1]:.5f}", flush=True)
