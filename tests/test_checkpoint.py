"""On-disk formats (section 8f rank 2): safetensors reader vs the `safetensors` package on the same file; config.json schema;
MLX affine de-quantisation on the GPU vs the documented formula (oracle/quant.py); a quantised micro checkpoint loaded end to end."""
import json
import os

import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S


def _write_ckpt(tmp_path, tensors, cfg=None):
    from safetensors.numpy import save_file
    p = os.path.join(tmp_path, "model.safetensors")
    save_file({k: np.ascontiguousarray(v) for k, v in tensors.items()}, p)
    if cfg is not None:
        json.dump(cfg, open(os.path.join(tmp_path, "config.json"), "w"))
    return p


def test_safetensors_reader_matches_library(tmp_path):
    from safetensors.numpy import load_file
    from mlx_swift_audio_amd import checkpoint as CK
    rng = np.random.default_rng(0)
    t = {"a.weight": rng.standard_normal((5, 7)).astype(np.float32), "b": rng.standard_normal(9).astype(np.float16),
         "q.weight": rng.integers(0, 2**32, (4, 8), dtype=np.uint32), "i": np.arange(6, dtype=np.int32).reshape(2, 3), "s": np.float32(3.5).reshape(())}
    p = _write_ckpt(str(tmp_path), t)
    got, want = CK.read_safetensors(p), load_file(p)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape
        np.testing.assert_array_equal(np.asarray(got[k]), want[k])


def test_model_dimensions_from_config(tmp_path):
    from mlx_swift_audio_amd import checkpoint as CK
    d = S.DIMS["large-v3-turbo"]
    cfg = {k: getattr(d, k) for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state",
                                      "n_text_head", "n_text_layer")}
    cfg["extra"] = "ignored"
    p = os.path.join(str(tmp_path), "config.json")
    json.dump(cfg, open(p, "w"))
    assert CK.load_model_dimensions(p) == d
    import mlx_swift_audio_amd as M
    cfg.pop("n_mels")
    json.dump(cfg, open(p, "w"))
    with pytest.raises(M.MiaError):
        CK.load_model_dimensions(p)


def test_quantize_round_trip_oracle():
    from oracle import quant as OQ
    rng = np.random.default_rng(1)
    w = rng.standard_normal((6, 256)).astype(np.float32)
    for bits in (4, 8):
        q, s, b = OQ.quantize_affine(w, 64, bits)
        assert q.shape == (6, 256 * bits // 32) and s.shape == b.shape == (6, 4)
        back = OQ.dequantize_affine(q, s, b, 64, bits)
        step = (w.reshape(6, 4, 64).max(-1) - w.reshape(6, 4, 64).min(-1)) / ((1 << bits) - 1)
        assert np.abs(back - w).max() <= 0.5 * step.max() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("sdt", ["f32", "f16"])
def test_dequant_gpu_matches_oracle(ctx, bits, sdt):
    from mlx_swift_audio_amd import checkpoint as CK
    from oracle import quant as OQ
    rng = np.random.default_rng(bits)
    w = rng.standard_normal((37, 320)).astype(np.float32)
    q, s, b = OQ.quantize_affine(w, 64, bits)
    if sdt == "f16":
        s, b = s.astype(np.float16), b.astype(np.float16)
    want = OQ.dequantize_affine(q, s.astype(np.float32), b.astype(np.float32), 64, bits)
    got = CK.dequantize_affine(ctx, q, s, b, 64, bits)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)       # one fma vs mul + add


@pytest.mark.gpu
def test_quantized_whisper_checkpoint_end_to_end(ctx, tmp_path):
    """A 4-bit micro checkpoint written the way mlx-community ships them loads and decodes like its de-quantised fp32 twin."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import checkpoint as CK, whisper as HW
    from oracle import quant as OQ, whisper as OW
    dims = OW.DIMS["micro.en"]
    w = OW.synthetic_weights(dims, seed=5)
    disk, dense = {}, {}
    for k, v in w.items():
        if k.endswith(".weight") and v.ndim == 2 and v.shape[1] % 64 == 0 and "embedding" not in k:
            q, s, b = OQ.quantize_affine(v, 64, 4)
            disk[k], disk[k[:-7] + ".scales"], disk[k[:-7] + ".biases"] = q, s.astype(np.float16), b.astype(np.float16)
            dense[k] = OQ.dequantize_affine(q, s.astype(np.float16).astype(np.float32), b.astype(np.float16).astype(np.float32), 64, 4)
        else:
            disk[k], dense[k] = v.astype(np.float32), v.astype(np.float32)
    cfg = {k: getattr(dims, k) for k in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer", "n_vocab", "n_text_ctx", "n_text_state",
                                         "n_text_head", "n_text_layer")}
    _write_ckpt(str(tmp_path), disk, cfg)
    d2, tensors = CK.load_whisper_checkpoint(ctx, str(tmp_path))
    assert d2 == dims and set(tensors) == set(dense)
    for k in dense:
        np.testing.assert_allclose(tensors[k], dense[k], rtol=1e-6, atol=1e-6)
    model = HW.WhisperModel.load(ctx, d2, tensors, m.F16)
    ref = HW.WhisperModel.load(ctx, dims, dense, m.F16)
    mel = (0.5 * np.random.default_rng(2).standard_normal((2, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    o = HW.DecodingOptions(suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220], max_new_tokens=12)
    a, b = HW.GreedyDecoder(model, o).decode(mel), HW.GreedyDecoder(ref, o).decode(mel)
    for x, y in zip(a, b):
        assert x.tokens == y.tokens
    model.close(); ref.close()
