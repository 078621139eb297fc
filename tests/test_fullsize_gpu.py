"""Full-size checks at BASELINE.json's shapes (random-init weights of the real architectures), beyond the micro-model parity suites:

* Whisper large-v3-turbo (the headline configuration): one 30 s clip through log-mel -> 32-layer encoder -> greedy decode, HIP vs the
  fp32 CPU oracle at full size (encoder features, first generated tokens, avg_logprob), plus the size-independent properties the
  bench relies on: a clip's tokens do not depend on which batch it sits in, and repeated runs are bit-identical.
* Qwen2-0.5B (CosyVoice2's LM backbone, 24 layers): last-position logits vs the oracle, batched prompt pass vs stepping.

These take ~1 minute (checkpoint generation + a 4 s CPU encoder pass), hence one test each."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def turbo(ctx):
    """One large-v3-turbo checkpoint (bf16-rounded, shared by the HIP model and the oracle) for the tests of this module."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import whisper as HW
    dims = S.DIMS["large-v3-turbo"]
    weights = S.synthetic_weights(dims, seed=0, style="survey", round_to="bf16")
    model = HW.WhisperModel.load(ctx, dims, weights, m.BF16)
    yield dims, weights, model
    model.close()


def test_whisper_large_v3_turbo_full_size(ctx, turbo):
    from mlx_swift_audio_amd import whisper as HW
    from oracle import logmel as OL
    from oracle import whisper as OW
    dims, weights, model = turbo
    n_new = 12
    o = HW.DecodingOptions(suppress_ids=S.synthetic_suppress_list(model.special), blank_ids=[220], max_new_tokens=n_new)
    clips = [S.synth_clip(i) for i in range(4)]
    solo = model.transcribe_windows(clips[:1], o)[0]
    feats = model.audio_features()[0]                                  # [1500, 1280]
    # ---- parity at full size against the CPU restatement (bf16-rounded weights and mel, fp32 arithmetic)
    ora = OW.WhisperOracle(dims, weights)
    mel = OW.round_array(OL.whisper_log_mel_spectrogram(clips[0], dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES], "bf16")[None]
    xa = ora.encode(mel)
    ref_feats = xa.numpy()[0]
    scale = np.abs(ref_feats).max()
    assert np.abs(feats - ref_feats).max() <= 0.03 * scale, (np.abs(feats - ref_feats).max(), scale)   # 32 bf16 layers, tolerance of test_whisper_gpu.py
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    ref = OW.greedy_decode(ora, st, xa, OW.DecodingOptions(suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220], max_new_tokens=n_new))
    # tokens agree up to the first step whose top-2 logit margin is within bf16 noise (the oracle records every step's margin)
    k = next((i for i, (a, b) in enumerate(zip(solo.tokens, ref.tokens)) if a != b), min(len(solo.tokens), len(ref.tokens)))
    assert k == len(ref.tokens) or ref.margins[k] < 0.05, (solo.tokens, ref.tokens, ref.margins)
    if k == len(ref.tokens):                        # (NaN on both sides when a step had every token masked: the reference's own arithmetic)
        assert (np.isnan(solo.avg_logprob) and np.isnan(ref.avg_logprob)) or abs(solo.avg_logprob - ref.avg_logprob) <= 0.02 * max(1.0, abs(ref.avg_logprob))
    # ---- size-independent properties
    batch = model.transcribe_windows(clips, o)
    assert batch[0].tokens == solo.tokens                                                     # batch invariance (fixed-order sums)
    assert np.array_equal(np.float32(batch[0].avg_logprob), np.float32(solo.avg_logprob), equal_nan=True)
    again = model.transcribe_windows(clips, o)
    assert [r.tokens for r in again] == [r.tokens for r in batch]                             # determinism
    assert all(len(r.tokens) == n_new for r in batch)                                          # random weights never emit EOT early


def test_whisper_turbo_headline_batch_32(ctx, turbo):
    """The configuration bench.py times, on the checkpoint bench.py times: 32 x 30 s clips in ONE batch (M = 48 000 rows -> the 256^2
    8-phase GEMM tile, the 32-row skinny decode tile), decoded to the FULL budget the bench runs (max_tokens 448 -> 445 generated
    tokens, 447 decoder steps: self-KV rows up to 447, the end of the positional table, 55 eight-step graph replays + 7 single steps).
    Two of the 32 clips are traced: the step graph's logits at EVERY one of the 447 positions against the oracle's teacher-forced
    logits on the same tokens and the same audio features, and the head's decisions replayed exactly on the traced logits
    (tests/_whisper_trace.py) -- so the comparison does not end at the first fork.  Encoder features of the two clips against the
    full-size oracle encoder; the oracle's free run for 64 steps, a fork from it legal only where the two measured logit errors cover
    its margin.  All 32 clips through batch invariance against runs of 4 clips (M = 6 000 rows -> the 128^2 tile): per-clip ids must be
    identical whatever batch, tile shape or row slot a clip has."""
    import torch
    from mlx_swift_audio_amd import whisper as HW
    from oracle import logmel as OL
    from oracle import whisper as OW
    from _whisper_trace import assert_fork_explained, check_clip, first_fork
    dims, weights, model = turbo
    sup = S.synthetic_suppress_list(model.special)
    o = HW.DecodingOptions(suppress_ids=sup, blank_ids=[220])          # nothing caps the run: the bench's 445 tokens per clip
    budget = 448 - 3
    clips = [S.synth_clip(i) for i in range(32)]
    traced = (5, 30)
    model.trace_logits(list(traced))
    big = model.transcribe_windows(clips, o)
    feats = model.audio_features()
    assert all(len(r.tokens) == budget for r in big)                     # random weights never emit EOT: 445 tokens, as the bench reports
    ora = OW.WhisperOracle(dims, weights)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    oo = OW.DecodingOptions(suppress_ids=sup, blank_ids=[220])
    for slot, b in enumerate(traced):
        mel = OW.round_array(OL.whisper_log_mel_spectrogram(clips[b], dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES], "bf16")[None]
        ref_feats = ora.encode(mel).numpy()[0]
        scale = np.abs(ref_feats).max()
        assert np.abs(feats[b] - ref_feats).max() <= 0.03 * scale, (b, np.abs(feats[b] - ref_feats).max(), scale)
        xa = torch.from_numpy(feats[b:b + 1])                           # the decoder alone: both sides attend the same features
        info = check_clip(model, ora, st, oo, big[b], slot, xa, "bf16", budget)
        assert info["n_pos"] == 447
        ref = OW.greedy_decode(ora, st, xa, OW.DecodingOptions(suppress_ids=sup, blank_ids=[220], max_new_tokens=64))
        k = first_fork(big[b].tokens[:64], ref.tokens)
        if k is not None:
            assert_fork_explained(info, ref, k)
    model.trace_logits([])
    # batch invariance.  The encoder's M-tile shape differs between the two runs (fp32 summation order inside a K-tile does not:
    # both tiles accumulate k in the same order), so ids must be IDENTICAL; encoder features are compared bit for bit too.
    for i in range(0, 32, 4):
        small = model.transcribe_windows(clips[i:i + 4], o)
        fs = model.audio_features()
        for j in range(4):
            assert small[j].tokens == big[i + j].tokens, (i + j,)
            assert np.array_equal(np.float32(small[j].avg_logprob), np.float32(big[i + j].avg_logprob), equal_nan=True)
        np.testing.assert_array_equal(fs, feats[i:i + 4])


def test_whisper_turbo_f16_ids_bit_exact(ctx):
    """large-v3-turbo at full size in f16 parity mode (the reference's storage type, WhisperSTT.swift:157,182) on a NON-degenerate
    checkpoint (style 'peaky', seed 18: picked offline with the CPU oracle among 50 seeds): the oracle's run of 2 clips x 64 tokens
    has >= 16 distinct ids per clip, finite avg_logprob, different ids per clip, and a smallest top-2 margin >= 10 x the logit noise
    measured here; HIP must emit exactly the oracle's ids -- whole run, end to end (log-mel, 32-layer encoder, decoder), no fork rule."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import whisper as HW
    from oracle import logmel as OL
    from oracle import whisper as OW
    from _whisper_trace import check_clip, first_fork, nondegenerate
    dims = S.DIMS["large-v3-turbo"]
    weights = S.synthetic_weights(dims, seed=18, style="peaky", round_to="f16")
    model = HW.WhisperModel.load(ctx, dims, weights, m.F16)
    n_new = 64
    sup = S.synthetic_suppress_list(model.special)
    kw = dict(suppress_ids=sup, blank_ids=[220], max_new_tokens=n_new)
    clips = [S.synth_clip(5), S.synth_clip(30)]
    model.trace_logits([0, 1])
    got = model.transcribe_windows(clips, HW.DecodingOptions(**kw))
    ora = OW.WhisperOracle(dims, weights)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    mel = np.stack([OW.round_array(OL.whisper_log_mel_spectrogram(c, dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES], "f16") for c in clips])
    xa = ora.encode(mel)
    oo = OW.DecodingOptions(**kw)
    refs = [OW.greedy_decode(ora, st, xa[b:b + 1], oo) for b in range(2)]
    min_margin = nondegenerate(refs, n_new)
    noise = 0.0
    for b in range(2):
        assert got[b].tokens == refs[b].tokens, (b, first_fork(got[b].tokens, refs[b].tokens), refs[b].margins)
        assert abs(got[b].avg_logprob - refs[b].avg_logprob) <= 5e-3
        noise = max(noise, check_clip(model, ora, st, oo, got[b], b, xa[b:b + 1], "f16", n_new, tol_scale=2.0)["noise_rms"])   # 32 encoder layers deep
    assert min_margin >= 10 * noise, (min_margin, noise)
    model.close()


def test_qwen2_half_billion_full_size(ctx):
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    from oracle import lm as OLM
    cfg = S.LM_CONFIGS["qwen2-0.5b"]
    w = S.lm_weights(cfg, seed=0, round_to="bf16")
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    ids = np.random.default_rng(0).integers(0, cfg.vocab, 24).tolist()
    got = model.forward(ids)                                           # 23 positions through the batched prompt pass + one step
    ref = OLM.LMOracle(cfg, w).forward(ids).numpy()[-1]
    assert np.abs(got - ref).max() <= 0.08 * ref.std(), (np.abs(got - ref).max(), ref.std())
    model.reset()
    for t in ids[:-1]:
        model.forward([t])
    stepped = model.forward([ids[-1]])
    assert np.abs(stepped - ref).max() <= 0.08 * ref.std()
    assert np.abs(got - stepped).max() <= 0.05 * ref.std()           # two fp32 summation orders in front of the same 16-bit roundings, 24 layers deep
    model.close()


def test_orpheus_3b_shape_two_layers(ctx):
    """Orpheus-3B geometry (d 3072, 24:8 GQA heads of 128, FFN 8192, V 156 940, Llama-3 RoPE scaling) at reduced depth (2 of 28
    layers): last-position logits of a 24-token prompt (batched prompt pass + one step) vs the fp32 oracle, stepping vs prompt pass,
    and a short sampled continuation checked token by token under the boundary rule."""
    import dataclasses
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    from oracle import lm as OLM
    cfg = dataclasses.replace(S.LM_CONFIGS["orpheus-3b"], n_layers=2)
    w = S.lm_weights(cfg, seed=1, round_to="bf16")
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    ora = OLM.LMOracle(cfg, w)
    ids = np.random.default_rng(1).integers(0, cfg.vocab, 24).tolist()
    got = model.forward(ids)
    ref = ora.forward(ids).numpy()[-1]
    assert got.shape == ref.shape == (cfg.vocab,)
    assert np.abs(got - ref).max() <= 0.08 * ref.std(), (np.abs(got - ref).max(), ref.std())
    model.reset()
    for t in ids[:-1]:
        model.forward([t])
    stepped = model.forward([ids[-1]])
    assert np.abs(stepped - ref).max() <= 0.08 * ref.std()
    n_new = 10
    u = np.random.default_rng(2).random(n_new).astype(np.float32)
    kw = {"temperature": 0.6, "top_p": 0.8, "rep_penalty": 1.3, "rep_window": 20, "max_new_tokens": n_new, "stop_ids": (128258,)}
    gen = model.generate(ids, u, temperature=0.6, top_p=0.8, rep_penalty=1.3, rep_window=20, max_new_tokens=n_new, stop_ids=(128258,))
    trace = []
    want = OLM.generate(ora, ids, kw, u, trace)
    k = next((i for i, (a, b) in enumerate(zip(gen, want)) if a != b), min(len(gen), len(want)))
    if not (k == len(gen) == len(want)):
        assert min(trace[k]) < 2e-2, (k, trace[k], gen, want)       # bf16: the CDF moves by ~max|logit error| / T
    model.close()


def test_orpheus_3b_vocabulary_head_on_packed_weights(ctx):
    """The packed-weight step at Orpheus-3B's real width and vocabulary (V 156 940: the head GEMM takes its own launch form, 4 tiles per
    one-wave workgroup over 9 809 tiles with a ragged last workgroup), one layer deep: logits of three stepped tokens against the 16-bit
    step on the de-quantised checkpoint (tolerance = the 16-bit copy's weight rounding, tests/test_lm_gpu.py) and against the fp32 oracle."""
    import dataclasses
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    from oracle import lm as OLM
    from test_lm_gpu import _quantized_checkpoint
    cfg = dataclasses.replace(S.LM_CONFIGS["orpheus-3b"], n_layers=1)
    packed, dense = _quantized_checkpoint(cfg, seed=4)
    model = HL.CausalLM.load(ctx, cfg, dense, m.BF16)
    model.attach_q4(packed)
    ids = np.random.default_rng(3).integers(0, cfg.vocab, 3).tolist()
    model.use_q4(False)
    want = [model.forward([t]).copy() for t in ids]
    model.use_q4(True)
    model.reset()
    ref = OLM.LMOracle(cfg, dense).forward(ids).numpy()
    for i, t in enumerate(ids):
        got = model.forward([t])
        sd = ref[i].std()
        assert got.shape == (cfg.vocab,) and np.isfinite(got).all()
        assert np.abs(got - want[i]).max() <= 0.06 * sd, (i, np.abs(got - want[i]).max(), sd)
        assert np.abs(got - ref[i]).max() <= 0.08 * sd, (i, np.abs(got - ref[i]).max(), sd)
    model.close()


def test_whisper_large_v3_full_depth(ctx):
    """large-v3 at FULL depth (32 encoder + 32 decoder layers, d 1280, 20 heads, V 51 866: the per-GPU model of BASELINE configs[4]),
    2 clips x 64 tokens, bf16 on the bench's N(0, 0.02^2) checkpoint style: encoder features of both clips against the full-size fp32
    oracle; the step graph's logits (291 kernel nodes per step: cross-KV layer strides up to layer 31) at all 66 positions against the
    oracle's teacher-forced logits on the same tokens and features, head decisions replayed exactly (tests/_whisper_trace.py).  With
    N(0, 0.02^2) weights the logits are nearly flat (std ~ 0.03), so the oracle's free run and HIP may split early (round 2 saw a
    split at generated index 2 on an 8-layer cut of this model): a split is legal only where the measured logit errors of the two
    tokens cover the oracle's margin -- asserted, and printed with the numbers when it happens -- for the oracle on HIP's features
    and for the oracle end to end (its own fp32 encoder output)."""
    import torch
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import whisper as HW
    from oracle import logmel as OL
    from oracle import whisper as OW
    from _whisper_trace import assert_fork_explained, check_clip, explain_fork_other_features, first_fork
    dims = S.DIMS["large-v3"]
    weights = S.synthetic_weights(dims, seed=3, style="survey", round_to="bf16")
    model = HW.WhisperModel.load(ctx, dims, weights, m.BF16)
    sup = S.synthetic_suppress_list(model.special)
    budget = 64
    o = HW.DecodingOptions(suppress_ids=sup, blank_ids=[220], max_new_tokens=budget)
    clips = [S.synth_clip(40), S.synth_clip(41)]
    model.trace_logits([0, 1])
    got = model.transcribe_windows(clips, o)
    feats = model.audio_features()
    ora = OW.WhisperOracle(dims, weights)
    del weights
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    oo = OW.DecodingOptions(suppress_ids=sup, blank_ids=[220], max_new_tokens=budget)
    for b in range(2):
        mel = OW.round_array(OL.whisper_log_mel_spectrogram(clips[b], dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES], "bf16")[None]
        xa_o = ora.encode(mel)
        scale = np.abs(xa_o.numpy()).max()
        assert np.abs(feats[b] - xa_o.numpy()[0]).max() <= 0.03 * scale
        xa = torch.from_numpy(feats[b:b + 1])
        info = check_clip(model, ora, st, oo, got[b], b, xa, "bf16", budget, tol_scale=2.0)      # 32 decoder layers deep (turbo: 4)
        assert info["n_pos"] == budget + 2
        ref = OW.greedy_decode(ora, st, xa, OW.DecodingOptions(suppress_ids=sup, blank_ids=[220], max_new_tokens=16))
        k = first_fork(got[b].tokens[:16], ref.tokens)
        if k is not None:
            bound = assert_fork_explained(info, ref, k)
            print(f"large-v3 clip {b}: HIP and the oracle split at generated index {k}: oracle margin {ref.margins[k]:.5f} <= measured "
                  f"|d| sum {bound:.5f} (logit std {info['ref'][info['n_init'] - 1 + k].std():.4f})")
        # the oracle end to end (its own fp32 encoder output): the comparison round 2 made
        ref2 = OW.greedy_decode(ora, st, xa_o, OW.DecodingOptions(suppress_ids=sup, blank_ids=[220], max_new_tokens=16))
        k2 = first_fork(got[b].tokens[:16], ref2.tokens)
        if k2 is not None:
            mg, da, dc, sd = explain_fork_other_features(info, ora, xa_o, ref2, k2)
            print(f"large-v3 clip {b}, oracle on its own features: split at generated index {k2}: oracle top-2 margin {mg:.5f}, measured "
                  f"logit error {da:.5f} (HIP's token) + {dc:.5f} (oracle's token), logit std {sd:.4f}")
        np.testing.assert_allclose(got[b].no_speech_prob, ref.no_speech_prob, rtol=0.1, atol=1e-6)
    model.close()


def test_orpheus_3b_full_depth(ctx):
    """Orpheus-3B at FULL depth (28 Llama-3 layers, d 3072, 24:8 GQA heads of 128, FFN 8192, V 156 940 -- BASELINE configs[2]'s
    backbone): last-position logits of a 24-token prompt (batched prompt pass + one step) and of a further stepped token against the
    fp32 oracle of the same bf16-rounded weights.  (Sampling / stop handling at this geometry: test_orpheus_3b_shape_two_layers; the
    packed 4- / 8-bit step: tests/test_lm_gpu.py.)"""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    from oracle import lm as OLM
    cfg = S.LM_CONFIGS["orpheus-3b"]
    w = S.lm_weights(cfg, seed=1, round_to="bf16")
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    ora = OLM.LMOracle(cfg, w)
    ids = np.random.default_rng(1).integers(0, cfg.vocab, 25).tolist()
    got = model.forward(ids[:24])
    got2 = model.forward(ids[24:])
    ref = ora.forward(ids).numpy()
    for g, r in ((got, ref[23]), (got2, ref[24])):
        assert g.shape == r.shape == (cfg.vocab,)
        assert np.abs(g - r).max() <= 0.12 * r.std(), (np.abs(g - r).max(), r.std())       # 28 bf16 layers (2 layers: 0.08)
    model.close()


def test_whisper_tiny_en_config0(ctx):
    """BASELINE configs[0]: Whisper tiny.en (80 mels, d 384, 6 heads, 4 + 4 layers, V 51 864, SOT sequence [sot]), greedy transcribe of
    10 s mono clips -- the whole model at its real size, f16 parity mode, on a NON-degenerate checkpoint (style 'peaky', seed picked
    offline with the oracle): the oracle's run has >= 16 distinct ids in 64, finite avg_logprob, the two clips differ, and its
    smallest top-2 margin is >= 10 x the measured logit noise; HIP must emit exactly the oracle's 64 ids per clip (no fork rule).
    Then one clip to the full 447-token budget with every position's logits against the oracle (tests/_whisper_trace.py)."""
    import torch
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import whisper as HW
    from oracle import logmel as OL
    from oracle import whisper as OW
    from _whisper_trace import check_clip, first_fork, nondegenerate
    dims = S.DIMS["tiny.en"]
    weights = S.synthetic_weights(dims, seed=46, style="peaky", round_to="f16")
    model = HW.WhisperModel.load(ctx, dims, weights, m.F16)
    n_new = 64
    sup = S.synthetic_suppress_list(model.special)
    kw = dict(suppress_ids=sup, blank_ids=[220], max_new_tokens=n_new)
    clips = [S.synth_clip(0, 160000), S.synth_clip(1, 160000)]        # 10 s
    model.trace_logits([0, 1])
    got = model.transcribe_windows(clips, HW.DecodingOptions(**kw))
    feats = model.audio_features()
    ora = OW.WhisperOracle(dims, weights)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    assert st.sot_sequence(0, "transcribe") == [st.sot]                  # English-only: WhisperTokenizer.swift:382-384
    mel = np.stack([OW.round_array(OL.whisper_log_mel_spectrogram(c, dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES], "f16") for c in clips])
    xa = ora.encode(mel)
    assert np.abs(feats - xa.numpy()).max() <= 0.01 * max(1.0, np.abs(xa.numpy()).max())
    oo = OW.DecodingOptions(**kw)
    refs = [OW.greedy_decode(ora, st, xa[b:b + 1], oo) for b in range(2)]
    min_margin = nondegenerate(refs, n_new)
    noise = 0.0
    for b in range(2):
        assert got[b].tokens == refs[b].tokens, (b, first_fork(got[b].tokens, refs[b].tokens), refs[b].margins)
        assert abs(got[b].avg_logprob - refs[b].avg_logprob) <= 3e-3
        noise = max(noise, check_clip(model, ora, st, oo, got[b], b, xa[b:b + 1], "f16", n_new)["noise_rms"])
    assert min_margin >= 10 * noise, (min_margin, noise)
    # the full budget: 447 generated tokens (or an earlier EOT), every position
    kw.pop("max_new_tokens")
    model.trace_logits([0])
    full = model.transcribe_windows(clips[:1], HW.DecodingOptions(**kw))[0]
    info = check_clip(model, ora, st, OW.DecodingOptions(**kw), full, 0, torch.from_numpy(model.audio_features()[0:1]), "f16", 447)
    assert info["n_pos"] >= 64
    model.close()
