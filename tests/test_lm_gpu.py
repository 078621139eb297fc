"""GPU parity: Llama-3 / Qwen2 decode blocks, the Orpheus sampler and the generation loop (HIP, through the C ABI) vs the fp32
CPU oracle on seeded random-init weights rounded to the storage type.

Tolerances: logits (O(1..10) magnitude) max |delta| <= 0.08 (bf16) / 0.015 (f16) relative to the logit std; the sampler is
integer-exact given the same logits and uniform; generated ids are exact unless the draw sits within 1e-4 of a CDF boundary
or a top-p boundary (asserted through the oracle's own CDF)."""
import numpy as np
import pytest

from mlx_swift_audio_amd import synthetic as S
from oracle import lm as OL

pytestmark = pytest.mark.gpu

SAMPLE_TOL = 5e-3    # f16 parity mode: |logit error| / T at these dims stays well below this


def _dt(name):
    import mlx_swift_audio_amd as m
    return m.BF16 if name == "bf16" else m.F16


@pytest.mark.parametrize("cfg_name", ["llama-micro", "llama-micro128", "qwen-micro"])
@pytest.mark.parametrize("dtype_name", ["bf16", "f16"])
def test_forward_logits_match_oracle(ctx, cfg_name, dtype_name):
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS[cfg_name]
    w = S.lm_weights(cfg, seed=2, round_to=dtype_name)
    model = HL.CausalLM.load(ctx, cfg, w, _dt(dtype_name))
    ora = OL.LMOracle(cfg, w)
    ids = [5, 17, 256, 999, 2048, 3, 42, 7, 7, 1500]
    ref = ora.forward(ids).numpy()
    got_last = model.forward(ids)
    scale = ref.std()
    tol = (0.08 if dtype_name == "bf16" else 0.015) * scale
    assert np.abs(got_last - ref[-1]).max() <= tol, (np.abs(got_last - ref[-1]).max(), tol)
    nxt = model.forward([11])                       # incremental step on the cached context
    ref2 = ora.forward([11]).numpy()[-1]
    assert np.abs(nxt - ref2).max() <= tol
    model.reset()
    again = model.forward(ids)
    np.testing.assert_array_equal(again, got_last)  # deterministic (fixed-order split-K, no atomics)
    model.close()


@pytest.mark.parametrize("cfg_name,n,max_ctx", [("llama-micro", 40, 256), ("llama-micro128", 131, 256), ("qwen-micro", 700, 1024)])
def test_batched_prompt_pass_matches_stepping(ctx, cfg_name, n, max_ctx):
    """forward(ids) ingests ids[:-1] through the batched prompt pass (MFMA GEMM rows, chunks of 512 positions); feeding the same
    ids one call at a time takes the single-row step graph for every position.  Same K/V rows up to fp32 summation order, so the
    last-position logits agree far inside the oracle tolerance, and both agree with the oracle."""
    import dataclasses
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = dataclasses.replace(S.LM_CONFIGS[cfg_name], max_ctx=max_ctx)
    w = S.lm_weights(cfg, seed=6, round_to="bf16")
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    ids = np.random.default_rng(n).integers(0, cfg.vocab, n).tolist()
    batched = model.forward(ids)
    nxt_b = model.forward([9])                      # a decode step on the K/V rows the prompt pass wrote
    model.reset()
    for t in ids[:-1]:
        model.forward([t])
    stepped = model.forward([ids[-1]])
    nxt_s = model.forward([9])
    scale = stepped.std()
    assert np.abs(batched - stepped).max() <= 0.02 * scale, np.abs(batched - stepped).max() / scale
    assert np.abs(nxt_b - nxt_s).max() <= 0.02 * scale
    if n <= 200:
        ref = OL.LMOracle(cfg, w).forward(ids).numpy()[-1]
        assert np.abs(batched - ref).max() <= 0.08 * ref.std()
    model.close()


def test_sampler_matches_oracle(ctx):
    from mlx_swift_audio_amd import lm as HL
    rng = np.random.default_rng(0)
    V = 20000
    mismatches = 0
    for trial in range(40):
        logits = (rng.standard_normal(V) * rng.choice([0.5, 2.0, 6.0])).astype(np.float32)
        hist = rng.integers(0, V, rng.integers(0, 21)).tolist()
        temp, top_p = float(rng.choice([0.6, 1.0])), float(rng.choice([0.8, 0.95, 0.3]))
        u = float(rng.random())
        f = OL.top_p_filter(logits, hist, 1.3, temp, top_p)
        ref = OL.sample_with_uniform(f, u)
        got = HL.sample_next_token(ctx, logits, hist, u, temp, top_p, 1.3)
        assert np.isfinite(f[got]), f"trial {trial}: token {got} is outside the reference's top-p set"
        if got != ref:
            # only legal when u sits on a CDF boundary (fp32 partial sums vs float64 cumsum)
            p = np.exp(f.astype(np.float64) - f[np.isfinite(f)].max()); p[~np.isfinite(f)] = 0
            c = np.cumsum(p) / p.sum()
            lo = c[got - 1] if got > 0 else 0.0
            assert abs(u - lo) < 1e-4 or abs(u - c[got]) < 1e-4, (trial, got, ref, u, lo, c[got])
            mismatches += 1
    assert mismatches <= 2


def test_top_p_kept_set_size(ctx):
    """The kept set must equal the reference's (first token crossing p included): probe it by sweeping u."""
    from mlx_swift_audio_amd import lm as HL
    logits = np.log(np.array([0.05, 0.5, 0.15, 0.3], np.float32))
    seen = {HL.sample_next_token(ctx, logits, [], u, 1.0, 0.7, 1.0) for u in np.linspace(0.001, 0.999, 97)}
    assert seen == {1, 3}                            # 0.5 + 0.3 crosses 0.7; 0.15 and 0.05 are dropped
    seen = {HL.sample_next_token(ctx, logits, [], u, 1.0, 0.4, 1.0) for u in np.linspace(0.001, 0.999, 23)}
    assert seen == {1}


def test_top_p_exact_ties_across_vocabulary_slices(ctx):
    """Planted ties: four distinct logit values over 9 000 tokens, so the top-p threshold falls INSIDE a group of bit-equal
    probabilities that spans several of the sampler's 32 vocabulary slices.  The kept part of that group must be its lowest indices
    (the reference's stable descending sort, OrpheusTTS.swift:417-455): every draw lies in the oracle's kept set and equals the oracle's
    inverse-CDF draw (a kept set off by one tie would shift the CDF by ~1e-3, ten times the boundary tolerance)."""
    from mlx_swift_audio_amd import lm as HL
    rng = np.random.default_rng(12)
    V = 9000
    logits = rng.choice(np.array([0.0, 0.7, 1.9, 3.1], np.float32), V, p=[0.55, 0.3, 0.1, 0.05]).astype(np.float32)
    partial_groups = 0
    for top_p in (0.35, 0.6, 0.9):
        f = OL.top_p_filter(logits, [], 1.0, 1.0, top_p)
        kept = np.isfinite(f)
        v = logits[kept].min()
        grp = np.flatnonzero(logits == v)
        n_kept = int(kept[grp].sum())
        assert kept[grp[:n_kept]].all() and not kept[grp[n_kept:]].any()       # the oracle keeps a prefix of the tie group
        partial_groups += 0 < n_kept < grp.size
        p = np.exp(f.astype(np.float64) - f[kept].max()); p[~kept] = 0
        c = np.cumsum(p) / p.sum()
        us = np.concatenate([np.linspace(0.0005, 0.9995, 40), [c[grp[n_kept - 1]] - 1e-5, 1.0 - 1e-6]])    # incl. the last kept tie
        for u in us:
            got = HL.sample_next_token(ctx, logits, [], float(u), 1.0, top_p, 1.0)
            assert kept[got], (top_p, u, got)
            ref = OL.sample_with_uniform(f, float(u))
            if got != ref:
                lo = c[got - 1] if got > 0 else 0.0
                assert abs(u - lo) < 1e-4 or abs(u - c[got]) < 1e-4, (top_p, u, got, ref)
    assert partial_groups >= 2


@pytest.mark.parametrize("cfg_name", ["llama-micro", "qwen-micro"])
def test_generate_matches_oracle(ctx, cfg_name):
    from mlx_swift_audio_amd import lm as HL
    import mlx_swift_audio_amd as m
    cfg = S.LM_CONFIGS[cfg_name]
    w = S.lm_weights(cfg, seed=4, round_to="f16")
    model = HL.CausalLM.load(ctx, cfg, w, m.F16)
    ora = OL.LMOracle(cfg, w)
    prompt = [100, 200, 300, 400, 500, 600]
    n_new = 24
    u = np.random.default_rng(3).random(n_new).astype(np.float32)
    stop = 2999
    got = model.generate(prompt, u, temperature=0.6, top_p=0.8, rep_penalty=1.3, rep_window=20, max_new_tokens=n_new, stop_ids=(stop,))
    trace = []
    ref = OL.generate(ora, prompt, {"temperature": 0.6, "top_p": 0.8, "rep_penalty": 1.3, "rep_window": 20, "max_new_tokens": n_new, "stop_ids": (stop,)}, u, trace)
    # a first divergence is legal only at a step where the ORACLE's own draw is within the f16 logit noise of flipping: the uniform
    # within SAMPLE_TOL of an edge of the chosen token's CDF interval, or a sorted cumulative probability within SAMPLE_TOL of top_p
    k = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
    if not (k == len(got) == len(ref)):
        cdf_d, topp_d = trace[k]
        assert min(cdf_d, topp_d) < SAMPLE_TOL, (k, cdf_d, topp_d, got, ref)
    model.close()


@pytest.mark.parametrize("cfg_name", ["llama-micro128", "qwen-micro"])
def test_generate_batch_equals_single_sequence_runs(ctx, cfg_name):
    """Sentence-level batching: n prompts of different lengths decoded side by side (rows of the same skinny GEMMs, own K/V cache,
    repetition window, uniforms and stop state each) give, sequence by sequence, exactly the ids of n separate generate() calls on the
    same handle capacity (the capacity picks the step's kernel chain -- <= 4: RMSNorm carried across the GEMMs, > 4: split-K + reduce --
    so within one capacity a sequence's ids never depend on its batch; across the two chains they agree to fp32 summation order, which
    the last check states through the oracle's boundary rule)."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS[cfg_name]
    w = S.lm_weights(cfg, seed=8, round_to="bf16")
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    rng = np.random.default_rng(5)
    prompts = [rng.integers(0, cfg.vocab - 1, n).tolist() for n in (3, 40, 1, 17, 9, 64)]      # incl. a one-token prompt (no prompt-pass rows)
    n_new = 40
    u = rng.random((len(prompts), n_new)).astype(np.float32)
    stop = int(rng.integers(0, cfg.vocab))                 # some sequences stop early, the others run to max_new_tokens
    kw = dict(temperature=0.8, top_p=0.9, rep_penalty=1.2, rep_window=16, max_new_tokens=n_new, stop_ids=(stop,))
    model.set_batch(8)
    solo = [model.generate(p, u[b], **kw) for b, p in enumerate(prompts)]   # one sequence at a time on the capacity-8 handle
    both = model.generate_batch(prompts, u, **kw)
    assert both == solo
    assert model.generate_batch(prompts[:2], u[:2], **kw) == solo[:2]      # a smaller batch on the same state
    with pytest.raises(m.MiaError):
        model.generate_batch(prompts * 2, np.concatenate([u, u]), **kw)   # 12 sequences > set_batch(8)
    # the low-latency chain (capacity <= 4): again batch == solo, bit for bit
    model.set_batch(4)
    solo4 = [model.generate(p, u[b], **kw) for b, p in enumerate(prompts[:4])]
    assert model.generate_batch(prompts[:4], u[:4], **kw) == solo4
    model.set_batch(1)
    assert model.generate(prompts[2], u[2], **kw) == solo4[2]
    # the two chains against the fp32 oracle under the sampling boundary rule (they differ from each other by fp32 summation order only)
    ora = OL.LMOracle(cfg, w)
    for b in (1, 3):
        for got in (solo[b], solo4[b]):
            _sampled_ids_follow_oracle(got, ora, prompts[b], kw, u[b], 2e-2)
    model.close()


def test_parse_output_host_logic():
    from mlx_swift_audio_amd import lm as HL
    off = HL.CODE_OFFSET
    frame = [off + 7, off + 4096 + 1, off + 2 * 4096 + 2, off + 3 * 4096 + 3, off + 4 * 4096 + 4, off + 5 * 4096 + 5, off + 6 * 4096 + 6]
    toks = [HL.AUDIO_CODE_DATA_START_MARKER] + frame * 2 + [HL.END_TOKEN]
    assert HL.parse_output(toks) == OL.parse_output(toks) == [[7, 7], [1, 4, 1, 4], [2, 3, 5, 6, 2, 3, 5, 6]]


def test_qwen2lm_ras_inference_matches_oracle(ctx):
    """CosyVoice2 Qwen2LM.inference: embedding-row prompt, speech-embedding feedback, llm_decoder head, RAS sampling with an
    explicit uniform stream (EOS rejection below min_len consumes extra draws on both sides)."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS["qwen-micro"]
    S_TOK = 200                                       # reduced speech vocabulary: EOS hits are likely -> exercises rejection + stop
    w = S.lm_weights(cfg, seed=5, round_to="f16")
    w.update(S.qwen2lm_extra_weights(cfg, S_TOK, seed=5, round_to="f16"))
    model = HL.CausalLM.load(ctx, cfg, w, m.F16)
    q = HL.Qwen2LM(model, w, speech_token_size=S_TOK)
    ora = OL.LMOracle(cfg, {k: v for k, v in w.items() if k.startswith("model.")})
    text, ptext, pspeech = [11, 12, 13, 14, 15, 16], [5, 6, 7], [3, 4, 5, 6]
    u = np.random.default_rng(9).random(4000).astype(np.float32)
    got = q.inference(text, ptext, pspeech, u)
    x = q.lm_input(text, ptext, pspeech)
    ref = OL.qwen2lm_inference(ora, x, w["llm_decoder.weight"], w["llm_decoder.bias"], w["speech_embedding.weight"], S_TOK,
                               int(len(text) * 2.0), int(len(text) * 20.0), u)
    assert all(0 <= t < S_TOK for t in got) and len(got) >= int(len(text) * 2.0) - 1
    k = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
    assert k >= 6, (got, ref)                         # streams may fork at an f16-moved CDF / top-k boundary, never early
    model.close()


def test_ras_batch_equals_single_utterance_runs(ctx):
    """CosyVoice2 utterance-level batching: embedding-row prompts of different lengths, per-utterance (min_len, max_len) and uniform
    streams (EOS rejections consume extra draws per utterance) -- ids identical to separate generate_ras calls."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS["qwen-micro"]
    S_TOK = 200
    w = S.lm_weights(cfg, seed=5, round_to="f16")
    w.update(S.qwen2lm_extra_weights(cfg, S_TOK, seed=5, round_to="f16"))
    model = HL.CausalLM.load(ctx, cfg, w, m.F16)
    rng = np.random.default_rng(11)
    lens = (15, 4, 33, 9)
    xs = [rng.standard_normal((n, cfg.hidden)).astype(np.float32) for n in lens]
    mins, maxs = [12, 3, 20, 8], [60, 25, 90, 40]
    u = rng.random((len(xs), 600)).astype(np.float32)
    solo = [model.generate_ras(xs[b], u[b], mins[b], maxs[b], S_TOK) for b in range(len(xs))]
    assert all(mins[b] - 1 <= len(solo[b]) <= maxs[b] for b in range(len(xs)))
    model.set_batch(4)
    assert model.generate_ras_batch(xs, u, mins, maxs, S_TOK) == solo
    assert model.generate_ras(xs[2], u[2], mins[2], maxs[2], S_TOK) == solo[2]
    model.close()


def test_orpheus_sentence_loop_batched_equals_sequential(ctx):
    """OrpheusTTS mirror: generate_chunks (sentences side by side) returns, per sentence, what generate_chunk returns.  (With a
    3 000-id micro vocabulary no id falls in Orpheus' audio-code range, so the SNAC leg yields empty audio on both sides; the SNAC
    decoder itself is covered by test_codec_gpu.py and parse_output by test_parse_output_host_logic.)"""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import codec as HC, lm as HL
    cfg = S.LM_CONFIGS["llama-micro"]
    w = S.lm_weights(cfg, seed=9, round_to="bf16")
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    scfg = S.SNAC_CONFIGS["snac_micro"]
    snac = HC.SNACDecoder.load(ctx, scfg, S.snac_weights(scfg, 1))
    tts = HL.OrpheusTTS(model, snac)
    rng = np.random.default_rng(2)
    sents = [rng.integers(0, cfg.vocab, n).tolist() for n in (12, 30, 5)]
    u = rng.random((3, 24)).astype(np.float32)
    seq = [tts.generate_chunk(sents[b], u[b], max_new_tokens=24) for b in range(3)]
    model.set_batch(3)
    par = tts.generate_chunks(sents, u, max_new_tokens=24)
    assert [g for g, _ in par] == [g for g, _ in seq]
    assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(par, seq))
    snac.close()
    model.close()


@pytest.mark.parametrize("n_tied", [5, 40])
def test_ras_top_k_with_exact_ties(ctx, n_tied):
    """The RAS sampler's top-k is a radix select with an ordered fallback when logits EQUAL to the k-th largest outnumber the slots left
    for them.  A zero llm_decoder weight makes every step's logits equal its bias, so exact ties can be planted: 5 tied values inside
    the top 25 (select path), 40 tied values straddling rank 25 (fallback path: the 15 lowest ids among them are kept).  Token streams
    must equal the oracle's stable-sort definition."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS["qwen-micro"]
    S_TOK = 200
    w = S.lm_weights(cfg, seed=5, round_to="f16")
    w.update(S.qwen2lm_extra_weights(cfg, S_TOK, seed=5, round_to="f16"))
    rng = np.random.default_rng(n_tied)
    bias = (rng.standard_normal(S_TOK + 3) * 0.5 - 4.0).astype(np.float32)
    ids = rng.permutation(S_TOK)                        # eos (= S_TOK) and the fill ids above it stay unlikely
    bias[ids[:10]] = np.linspace(3.0, 2.1, 10, dtype=np.float32)
    bias[ids[10:10 + n_tied]] = np.float32(1.5)
    w["llm_decoder.weight"] = np.zeros_like(w["llm_decoder.weight"])
    w["llm_decoder.bias"] = bias
    model = HL.CausalLM.load(ctx, cfg, w, m.F16)
    x = rng.standard_normal((6, cfg.hidden)).astype(np.float32)
    u = rng.random(400).astype(np.float32)
    n = 60
    got = model.generate_ras(x, u, 0, n, S_TOK)
    logp = bias.astype(np.float64) - np.log(np.exp(bias.astype(np.float64)).sum())
    draws = iter(list(u) + [u[-1]] * 1000)
    want = []
    for _ in range(n):
        t = OL.ras_sampling(logp, want, draws)
        if t == S_TOK:
            break
        if t < S_TOK:
            want.append(t)
    assert got == want
    model.close()


def _quantized_checkpoint(cfg, seed, scale_dtype=np.float16, bits=4):
    """A q4 / q8 group-64 checkpoint in the MLX layout plus its de-quantised tensors: every step Linear (and the tied embedding) of a
    seeded random-init model is quantised with the test-side restatement of mx.quantize (oracle/quant.py), scales / biases rounded to
    the 16-bit type a checkpoint stores them in."""
    from oracle import quant as OQ
    w = S.lm_weights(cfg, seed=seed)
    packed, dense = {}, dict(w)
    names = ["model.embed_tokens"] if cfg.tie_embeddings else ["lm_head"]
    for l in range(cfg.n_layers):
        p = f"model.layers.{l}"
        names += [p + ".self_attn." + n + "_proj" for n in "qkvo"] + [p + ".mlp." + n + "_proj" for n in ("gate", "up", "down")]
    for n in names:
        pk, sc, bi = OQ.quantize_affine(w[n + ".weight"], 64, bits)
        sc, bi = sc.astype(scale_dtype), bi.astype(scale_dtype)
        packed[n + ".weight"], packed[n + ".scales"], packed[n + ".biases"] = pk, sc, bi
        dense[n + ".weight"] = OQ.dequantize_affine(pk, sc.astype(np.float32), bi.astype(np.float32), 64, bits)
    return packed, dense


def _sampled_ids_follow_oracle(gen, ora, prompt, kw, u, tol):
    """gen == the oracle's run on the same weights, or the first difference sits at a draw whose uniform is within `tol` of a CDF edge /
    the top-p cut in the ORACLE (oracle/lm.py:sample_boundary_distance); after a proven fork the comparison stops."""
    trace = []
    want = OL.generate(ora, prompt, dict(kw), u, trace)
    k = next((i for i, (a, b) in enumerate(zip(gen, want)) if a != b), min(len(gen), len(want)))
    if not (k == len(gen) == len(want)):
        assert min(trace[k]) < tol, (k, trace[k], gen, want)


@pytest.mark.parametrize("cfg_name,dtype_name,bits", [("llama-micro128", "bf16", 4), ("qwen-micro", "f16", 4), ("llama-micro128", "bf16", 8), ("qwen-micro", "f16", 8)])
def test_packed_step_matches_oracle_and_expanded_checkpoint(ctx, cfg_name, dtype_name, bits):
    """Packed MLX-affine 4- / 8-bit step (mia_lm_attach_quantized: the codes go through the MFMA as exact 16-bit integers, scale and bias
    are applied per 64-input group to fp32 group sums -- MLX's own qmv arithmetic) against (i) the fp32 oracle on the de-quantised
    weights, which is what that arithmetic computes up to 16-bit activations, and (ii) the 16-bit step of the SAME handle on the
    checkpoint expanded at load, which additionally rounds every de-quantised weight to 16 bit: the two must agree to that rounding
    (a few 1e-3 of the logit spread), step after step, and the packed step must sit at least as close to the oracle."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS[cfg_name]
    packed, dense = _quantized_checkpoint(cfg, seed=11, bits=bits)
    model = HL.CausalLM.load(ctx, cfg, dense, _dt(dtype_name))
    ids = np.random.default_rng(2).integers(0, cfg.vocab, 9).tolist()
    model.attach_q4(packed, bits=bits)
    with pytest.raises(m.MiaError):
        model.attach_q4(packed, bits=bits)          # one attach per handle
    model.use_q4(False)                             # the 16-bit weights of the same handle (same cross-workgroup splits)
    ref_steps = []
    for t in ids:                                   # token by token: every call is one step graph launch
        ref_steps.append(model.forward([t]).copy())
    model.use_q4(True)
    model.reset()
    q_steps = [model.forward([t]).copy() for t in ids]
    ora = OL.LMOracle(cfg, dense)                   # fp32 de-quantised weights, NOT rounded to 16 bit
    ref = ora.forward(ids).numpy()
    tol16 = 0.08 if dtype_name == "bf16" else 0.02
    for i, (got, want) in enumerate(zip(q_steps, ref_steps)):
        sd = ref[i].std()
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= (0.06 if dtype_name == "bf16" else 0.01) * sd, (i, np.abs(got - want).max(), sd)   # the expanded checkpoint's weight rounding
        assert np.abs(got - ref[i]).max() <= tol16 * sd, (i, np.abs(got - ref[i]).max(), sd)
    # sampled generation on the packed step follows the oracle under the boundary rule
    u = np.random.default_rng(4).random(20).astype(np.float32)
    kw = dict(temperature=0.7, top_p=0.9, rep_penalty=1.2, rep_window=16, max_new_tokens=20, stop_ids=(cfg.vocab - 1,))
    _sampled_ids_follow_oracle(model.generate(ids, u, **kw), ora, ids, kw, u, 2e-2 if dtype_name == "bf16" else 5e-3)
    # batched: sequence b of a batch equals its own single-sequence run (same kernels, rows side by side).  Prompts go token by token
    # through the packed step here: the batched prompt pass runs on the 16-bit copy (weights rounded), and whether a prompt takes it
    # depends on how many rows a call has -- a batch and a single run would otherwise differ by that rounding
    model.set_debug(2)
    model.set_batch(4)
    prompts = [ids, ids[:3], ids[2:], ids[::-1]]
    ub = np.random.default_rng(5).random((4, 12)).astype(np.float32)
    kw["max_new_tokens"] = 12
    qa = model.generate_batch(prompts, ub, **kw)
    model.set_batch(1)
    for b, pr in enumerate(prompts):
        assert model.generate(pr, ub[b], **kw) == qa[b], b
    model.close()
    # an incomplete tensor set is rejected and leaves the handle on its 16-bit weights
    fresh = HL.CausalLM.load(ctx, cfg, dense, _dt(dtype_name))
    with pytest.raises(m.MiaError):
        fresh.attach_q4({k: v for k, v in packed.items() if "q_proj" not in k}, bits=bits)
    with pytest.raises(m.MiaError):
        fresh.use_q4(True)
    assert np.array_equal(fresh.forward(ids[:1]), ref_steps[0])
    fresh.close()


@pytest.mark.parametrize("bits", [4, 8])
def test_packed_step_mid_size_batch_32(ctx, bits):
    """The packed step at a shape that exercises what the micro models cannot: several 128-input blocks per wave (the register ring),
    4-wave workgroups, cross-workgroup K splits, the 4-tile head and a FULL 32-row batch (both 16-row MFMA halves).  Logits against the
    16-bit step on the de-quantised weights and the fp32 oracle; 32 sequences side by side give, sequence by sequence, the ids of their
    own single-sequence runs."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import lm as HL
    cfg = S.LM_CONFIGS["llama-q4mid"]
    packed, dense = _quantized_checkpoint(cfg, seed=21, bits=bits)
    model = HL.CausalLM.load(ctx, cfg, dense, m.BF16)
    model.attach_q4(packed, bits=bits)
    rng = np.random.default_rng(6)
    ids = rng.integers(0, cfg.vocab, 12).tolist()
    model.use_q4(False)
    want = [model.forward([t]).copy() for t in ids]
    model.use_q4(True)
    model.reset()
    ref = OL.LMOracle(cfg, dense).forward(ids).numpy()
    for i, (t, w_) in enumerate(zip(ids, want)):
        got = model.forward([t])
        sd = ref[i].std()
        assert np.isfinite(got).all() and np.abs(got - w_).max() <= 0.06 * sd, (i, np.abs(got - w_).max(), sd)
        assert np.abs(got - ref[i]).max() <= 0.08 * sd, (i, np.abs(got - ref[i]).max(), sd)
    model.set_debug(2)              # prompts token by token through the packed step (see the test above)
    model.set_batch(32)
    prompts = [rng.integers(0, cfg.vocab, int(n)).tolist() for n in rng.integers(1, 20, 32)]
    ub = rng.random((32, 16)).astype(np.float32)
    kw = dict(temperature=0.8, top_p=0.9, rep_penalty=1.1, rep_window=8, max_new_tokens=16, stop_ids=(cfg.vocab - 1,))
    q = model.generate_batch(prompts, ub, **kw)
    assert all(0 <= t < cfg.vocab for seq in q for t in seq)
    for b in (0, 7, 31):            # rows of a 32-row batch (M > 16: both MFMA halves) equal the single-sequence (M16) kernel's ids
                                    # on the same capacity (same kernel chain; mia.h, mia_lm_generate_batch)
        assert model.generate(prompts[b], ub[b], **kw) == q[b], b
    model.close()
