"""Generates tests/golden/*.npz: small frozen input/output vectors of the hot path's stages.

PROVENANCE: the reference (Swift + MLX) cannot run in this image and ships no tensor-level golden vectors for this path
(SURVEY.md section 8c), so these vectors are produced by the CPU restatement in oracle/ -- they freeze the restatement (a drift
guard for both the oracle and the HIP path), they are NOT outputs of the reference.  "parity unpinned" still applies.

    python tests/golden/make_golden.py          # rewrites the fixtures (seeded, deterministic up to libm / BLAS rounding)

Inputs are stored next to the expected outputs so that the tests do not depend on RNG reproducibility; weights are NOT stored:
they are regenerated from mlx-swift-audio_amd/synthetic.py (PCG64 streams keyed by tensor name) and a checksum of a few tensors is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from mlx_swift_audio_amd import synthetic as S  # noqa: E402
from oracle import campplus as OCP, codec as OC, flow as OF, hift as OH, lm as OLM, logmel as OL, s3tok as OS, whisper as OW  # noqa: E402


def wsum(w, keys):
    return np.asarray([float(np.asarray(w[k], np.float64).sum()) for k in keys], np.float64)


def logmel():
    clip = OL.synth_clip(3, 16000)
    y24 = OL.synth_clip(4, 7200)
    np.savez_compressed(os.path.join(OUT, "logmel.npz"), clip=clip, whisper80=OL.whisper_log_mel_spectrogram(clip, 80),
                        whisper128=OL.whisper_log_mel_spectrogram(clip, 128), s3_128=OL.s3_log_mel_spectrogram(clip, 128),
                        y24=y24, s3gen80=OL.s3gen_mel_spectrogram(y24))


def whisper():
    """micro.en, f16 storage, the NON-degenerate 'peaky' checkpoint (seed 77: picked offline for the largest smallest-margin): 4 clips x
    64 greedy tokens with timestamps -- varied ids, finite avg_logprob -- plus a slice of the teacher-forced logits of clip 0
    (every 97th vocabulary entry at 5 positions) so that the step path's logits are pinned numerically, not only their argmax."""
    dims = OW.DIMS["micro.en"]
    w = OW.synthetic_weights(dims, seed=77, style="peaky", round_to="f16")
    ora = OW.WhisperOracle(dims, w)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    rng = np.random.default_rng(1)
    mel = OW.round_array((0.5 * rng.standard_normal((4, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32), "f16")
    xa = ora.encode(mel)
    oo = OW.DecodingOptions(timestamps=True, suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220, 50256 - 1], max_new_tokens=64)
    refs = [OW.greedy_decode(ora, st, xa[b:b + 1], oo) for b in range(4)]
    assert all(len(r.tokens) == 64 and len(set(r.tokens)) >= 16 and np.isfinite(r.avg_logprob) for r in refs)
    toks = np.asarray([r.tokens for r in refs], np.int32)
    init, _ = OW.initial_tokens(st, oo)
    pos = np.asarray([0, 1, 17, 40, 63], np.int32)
    logits = OW.teacher_forced_logits(ora, xa[0:1], init + refs[0].tokens)[pos][:, ::97]
    np.savez_compressed(os.path.join(OUT, "whisper_micro_en_f16.npz"), mel=mel, features=xa.numpy()[:, :8], tokens=toks,
                        margins=np.asarray([min(r.margins) for r in refs], np.float32),
                        avg_logprob=np.asarray([r.avg_logprob for r in refs], np.float32),
                        no_speech_prob=np.asarray([r.no_speech_prob for r in refs], np.float32),
                        logit_pos=pos, logits_clip0=logits.astype(np.float32),
                        wsum=wsum(w, ["encoder.conv1.weight", "decoder.token_embedding.weight", "decoder.ln.weight"]))


def codecs():
    cfg = S.SNAC_CONFIGS["snac_micro"]
    w = S.snac_weights(cfg, seed=3)
    rng = np.random.default_rng(5)
    n = 5
    codes = [rng.integers(0, cfg.codebook_size, n * (cfg.vq_strides[0] // s)).astype(np.int32) for s in cfg.vq_strides]
    ora = OC.SNACOracle(cfg, w)
    noise = rng.standard_normal(ora.noise_len(n * cfg.vq_strides[0])).astype(np.float32)
    d = {f"snac_codes{i}": c for i, c in enumerate(codes)}
    d.update(snac_noise=noise, snac_pcm=ora.decode([c.tolist() for c in codes], noise))
    dcfg = S.DAC_CONFIGS["dac_micro"]
    dw = S.dac_weights(dcfg, seed=4)
    dcodes = rng.integers(0, dcfg.codebook_size, (dcfg.n_codebooks, 9)).astype(np.int32)
    d.update(dac_codes=dcodes, dac_pcm=OC.DACOracle(dcfg, dw).decode_from_codes(dcodes))
    np.savez_compressed(os.path.join(OUT, "codecs_micro.npz"), **d)


def cosyvoice2():
    rng = np.random.default_rng(7)
    # S3 tokenizer
    scfg = S.S3_CONFIGS["s3_micro"]
    sw = S.s3_weights(scfg, 2)
    clip = OL.synth_clip(6, 16000 * 2)
    mel128 = OL.s3_log_mel_spectrogram(clip, scfg.n_mels)
    ids, n, _ = OS.S3Oracle(scfg, sw).quantize(mel128[None], np.asarray([mel128.shape[1]]))
    # flow
    fcfg = S.FLOW_CONFIGS["flow_micro"]
    fw = S.flow_weights(fcfg, seed=0)
    tok = rng.integers(0, fcfg.vocab_size, 7).astype(np.int32)
    ptok = rng.integers(0, fcfg.vocab_size, 3).astype(np.int32)
    pf = rng.standard_normal((6, 80)).astype(np.float32)
    emb = rng.standard_normal(fcfg.spk_embed_dim).astype(np.float32)
    z = rng.standard_normal((80, 20)).astype(np.float32)
    mel, _ = OF.inference(fw, fcfg, tok, ptok, pf, emb, z)
    # HiFT
    hcfg = S.HIFT_CONFIGS["hift_micro"]
    hw = S.hift_weights(hcfg, seed=0)
    hmel = (rng.standard_normal((80, 4)) * 1.5 - 2).astype(np.float32)
    f0 = OH.f0_predictor(hw, hmel)
    hnoise = rng.standard_normal((4 * 480, 9)).astype(np.float32)
    src = OH.source(hw, hcfg, f0, hnoise)
    pcm = OH.decode(hw, hcfg, hmel, src)
    np.savez_compressed(os.path.join(OUT, "cosyvoice2_micro.npz"), s3_mel128=mel128, s3_ids=ids[0, :n[0]].astype(np.int32),
                        flow_token=tok, flow_prompt_token=ptok, flow_prompt_feat=pf, flow_embedding=emb, flow_z=z, flow_mel=mel,
                        hift_mel=hmel, hift_f0=f0, hift_noise=hnoise, hift_source=src, hift_pcm=pcm)


def lm():
    cfg = S.LM_CONFIGS["llama-micro"]
    w = S.lm_weights(cfg, seed=1, round_to="f16")
    ora = OLM.LMOracle(cfg, w)
    ids = np.asarray([5, 17, 99, 3, 250, 7], np.int64)
    logits = ora.forward(ids)
    np.savez_compressed(os.path.join(OUT, "lm_llama_micro_f16.npz"), ids=ids.astype(np.int32), last_logits=np.asarray(logits[-1], np.float32),
                        wsum=wsum(w, ["model.embed_tokens.weight"]))


def campplus():
    w = S.campplus_weights(2)
    clip = OL.synth_clip(8, 16000)[:12000]
    fb = OCP.kaldi_fbank(clip)
    np.savez_compressed(os.path.join(OUT, "campplus.npz"), clip=clip, fbank=fb, embedding=OCP.CAMPPlusOracle(w).inference(clip)[0],
                        wsum=wsum(w, ["head.conv1.weight", "blocks.2.layers.15.cam_layer.linear_local.weight", "dense.linear.weight"]))


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (logmel, whisper, codecs, cosyvoice2, lm, campplus):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
