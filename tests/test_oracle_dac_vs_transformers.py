"""Independent pin of oracle/codec.py's DAC restatement (row a14): `transformers`' DacModel -- a separate implementation of the Descript
Audio Codec the Swift port mirrors (Codec/DAC/*.swift) -- loaded with the SAME weights (weight norm folded, MLX [out, k, in] layouts
permuted to torch's), must give the oracle's waveform from codes, and the oracle's codes from audio.  Codes are integers: equal, except
where the oracle's own best / second-best codebook distances tie within 1e-5 (asserted)."""
import numpy as np
import pytest
import torch

from mlx_swift_audio_amd import synthetic as S
from oracle import codec as OC

transformers = pytest.importorskip("transformers")
from transformers import DacConfig, DacModel  # noqa: E402


def _hf_model(cfg, w):
    hf = DacModel(DacConfig(encoder_hidden_size=cfg.encoder_dim, downsampling_ratios=list(cfg.encoder_rates), decoder_hidden_size=cfg.decoder_dim,
                            upsampling_ratios=list(cfg.decoder_rates), n_codebooks=cfg.n_codebooks, codebook_size=cfg.codebook_size,
                            codebook_dim=cfg.codebook_dim)).eval()
    W = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in w.items()}

    def conv(p):          # weight-normed Conv1d: g v / ||v|| over all but the output axis; MLX [O, K, I] -> torch [O, I, K]
        v, g = W[p + ".weight_v"], W[p + ".weight_g"]
        return (g * v / (OC._norm_except(v, 0) + 1e-12)).permute(0, 2, 1).contiguous(), W[p + ".bias"]

    def convt(p):         # weight-normed ConvTranspose1d: norm over all but the INPUT axis; MLX [O, K, I] -> torch [I, O, K]
        v, g = W[p + ".weight_v"], W[p + ".weight_g"]
        return (g * v / (OC._norm_except(v, 2) + 1e-12)).permute(2, 0, 1).contiguous(), W[p + ".bias"]

    sd = {}

    def put(name, wb):
        sd[name + ".weight"], sd[name + ".bias"] = wb

    def alpha(name, p):
        sd[name + ".alpha"] = W[p + ".alpha"].reshape(1, -1, 1)

    def res_units(dst, src):          # three DACResidualUnits: [snake, conv7 dilated, snake, conv1]
        for r in range(3):
            u = f"{src}{r}.block.layers." if dst.startswith("encoder") else f"{src}{2 + r}.block.layers."
            alpha(f"{dst}.res_unit{r + 1}.snake1", u + "0"); put(f"{dst}.res_unit{r + 1}.conv1", conv(u + "1"))
            alpha(f"{dst}.res_unit{r + 1}.snake2", u + "2"); put(f"{dst}.res_unit{r + 1}.conv2", conv(u + "3"))

    E = "encoder.block.layers."
    put("encoder.conv1", conv(E + "0"))
    for i in range(len(cfg.encoder_rates)):
        b = f"{E}{1 + i}.block.layers."
        res_units(f"encoder.block.{i}", b)
        alpha(f"encoder.block.{i}.snake1", b + "3"); put(f"encoder.block.{i}.conv1", conv(b + "4"))
    ne = len(cfg.encoder_rates)
    alpha("encoder.snake1", f"{E}{1 + ne}"); put("encoder.conv2", conv(f"{E}{2 + ne}"))
    P = "decoder.model.layers."
    put("decoder.conv1", conv(P + "0"))
    for i in range(len(cfg.decoder_rates)):
        b = f"{P}{1 + i}.block.layers."
        alpha(f"decoder.block.{i}.snake1", b + "0"); put(f"decoder.block.{i}.conv_t1", convt(b + "1"))
        res_units(f"decoder.block.{i}", b)
    nd = len(cfg.decoder_rates)
    alpha("decoder.snake1", f"{P}{1 + nd}"); put("decoder.conv2", conv(f"{P}{2 + nd}"))
    for q in range(cfg.n_codebooks):
        p = f"quantizer.quantizers.{q}"
        put(p + ".in_proj", conv(p + ".in_proj")); put(p + ".out_proj", conv(p + ".out_proj"))
        sd[p + ".codebook.weight"] = W[p + ".codebook.weight"]
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return hf


@pytest.mark.parametrize("name", ["dac_micro", "dac_speech"])
def test_dac_oracle_matches_transformers(name):
    cfg = S.DAC_CONFIGS[name]
    w = S.dac_weights(cfg, seed=3)
    ora, hf = OC.DACOracle(cfg, w), _hf_model(cfg, w)
    rng = np.random.default_rng(1)
    # ---- decoder: codes -> waveform
    T = 6 if name == "dac_speech" else 11
    codes = rng.integers(0, cfg.codebook_size, (cfg.n_codebooks, T))
    want = ora.decode_from_codes(codes)
    with torch.no_grad():
        got = hf.decode(audio_codes=torch.from_numpy(codes)[None]).audio_values.reshape(-1).numpy()
    assert got.shape == want.shape
    # (the real geometry with N(0, sigma) weights drives the pre-tanh signal to ~1e2: fp32 summation-order noise of the two conv
    # implementations is amplified accordingly; a structural difference would be O(1))
    np.testing.assert_allclose(got, want, atol=2e-5 if name == "dac_micro" else 3e-3, rtol=1e-4)
    # ---- encoder + residual VQ: audio -> codes (a whole number of hops: the HF model leaves padding to its feature extractor)
    hop = int(np.prod(cfg.encoder_rates))
    audio = (0.3 * rng.standard_normal((4 if name == "dac_speech" else 9) * hop)).astype(np.float32)
    ocodes, gaps = ora.encode(audio)
    with torch.no_grad():
        hcodes = hf.encode(torch.from_numpy(audio)[None, None]).audio_codes[0].numpy()
    assert hcodes.shape == ocodes.shape
    diff = hcodes != ocodes
    # a stage may differ only at an exact near-tie of the oracle's own two best entries; later stages of that frame then see another residual
    first = np.argmax(diff, axis=0)
    for t in np.flatnonzero(diff.any(axis=0)):
        assert gaps[first[t], t] <= 1e-5, (name, t, first[t], float(gaps[first[t], t]))
    assert diff.mean() <= 0.02
