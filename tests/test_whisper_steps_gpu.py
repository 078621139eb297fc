"""GPU parity of the Whisper decode STEP PATH, position by position (reduced dims; the full-size runs of the same checks are in
tests/test_fullsize_gpu.py).  See tests/_whisper_trace.py for what the trace check proves.

Checkpoints: synthetic style 'peaky' (mlx_swift_audio_amd/synthetic.py) -- random-init weights re-balanced so that greedy decoding is
not degenerate.  Every test first asserts that on the ORACLE's own run (>= 16 distinct ids in 64, finite avg_logprob, clips differ,
smallest top-2 margin >= 10 x the logit noise measured in the same test) and then demands the ids BIT-EXACT for the whole run, with
no fork rule.  The seeds were picked offline with the CPU oracle alone (largest smallest-margin among 320 seeds)."""
import numpy as np
import pytest
import torch

from oracle import whisper as OW

from _whisper_trace import assert_fork_explained, check_clip, first_fork, nondegenerate

pytestmark = pytest.mark.gpu


def _dt(name):
    import mlx_swift_audio_amd as m
    return m.BF16 if name == "bf16" else m.F16


def _setup(ctx, dims_name, dtype_name, seed, style="peaky"):
    from mlx_swift_audio_amd import whisper as HW
    dims = OW.DIMS[dims_name]
    weights = OW.synthetic_weights(dims, seed=seed, style=style, round_to=dtype_name)
    return dims, OW.WhisperOracle(dims, weights), HW.WhisperModel.load(ctx, dims, weights, _dt(dtype_name))


def _mel(dims, B, seed, dtype_name):
    rng = np.random.default_rng(seed)
    return OW.round_array((0.5 * rng.standard_normal((B, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32), dtype_name)


@pytest.mark.parametrize("dims_name,timestamps,seed", [("micro.en", True, 77), ("micro", False, 157)])
def test_ids_bit_exact_on_nondegenerate_checkpoint(ctx, dims_name, timestamps, seed):
    """f16 parity mode, 4 clips x 64 tokens: the oracle's run is varied and well separated; HIP must emit exactly its ids."""
    from mlx_swift_audio_amd import whisper as HW
    dims, ora, model = _setup(ctx, dims_name, "f16", seed)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    sup = OW.synthetic_suppress_list(st)
    B, n_new = 4, 64
    mel = _mel(dims, B, 1, "f16")
    kw = dict(timestamps=timestamps, suppress_ids=sup, blank_ids=[220, 50255], max_new_tokens=n_new)
    oo = OW.DecodingOptions(**kw)
    xa = ora.encode(mel)
    refs = [OW.greedy_decode(ora, st, xa[b:b + 1], oo) for b in range(B)]
    min_margin = nondegenerate(refs, n_new)
    model.trace_logits(list(range(B)))
    res = HW.GreedyDecoder(model, HW.DecodingOptions(**kw)).decode(mel)
    noise = 0.0
    for b in range(B):
        assert res[b].tokens == refs[b].tokens, (b, first_fork(res[b].tokens, refs[b].tokens), refs[b].margins)   # whole run, no fork rule
        assert abs(res[b].avg_logprob - refs[b].avg_logprob) <= 3e-3, (res[b].avg_logprob, refs[b].avg_logprob)
        np.testing.assert_allclose(res[b].no_speech_prob, refs[b].no_speech_prob, rtol=0.02, atol=1e-7)
        info = check_clip(model, ora, st, oo, res[b], b, xa[b:b + 1], "f16", n_new)     # end to end: the oracle's own features
        noise = max(noise, info["noise_rms"])
    # the margin the ids rest on, against the measured 16-bit noise of the logits
    assert min_margin >= 10 * noise, (min_margin, noise)
    model.close()


@pytest.mark.parametrize("dtype_name", ["f16", "bf16"])
def test_every_step_to_the_full_budget(ctx, dtype_name):
    """max_tokens = 448 with nothing capping the run: the step graph runs to the end of the positional table (self-KV rows up to 447,
    the 8-step graph replays plus the ragged single-step tail).  Every position's logits of 3 clips out of 5 against the oracle,
    the head's decisions replayed exactly; the same call with the captured graph switched off and with the one-workgroup head must
    give identical ids and bit-identical logits (the graph replays what direct launches compute)."""
    from mlx_swift_audio_amd import whisper as HW
    dims, ora, model = _setup(ctx, "micro.en", dtype_name, 77)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    sup = OW.synthetic_suppress_list(st)
    B = 5
    mel = _mel(dims, B, 4, dtype_name)
    kw = dict(timestamps=True, suppress_ids=sup, blank_ids=[220, 50255])
    oo = OW.DecodingOptions(**kw)
    budget = 448 - 1                                       # [sot] is the whole forced prefix of an English-only model
    traced = [0, 2, 4]
    model.trace_logits(traced)
    res = HW.GreedyDecoder(model, HW.DecodingOptions(**kw)).decode(mel)
    feats = model.audio_features()
    assert max(len(r.tokens) for r in res) == budget, [len(r.tokens) for r in res]       # at least one clip runs into the budget
    infos = []
    for slot, b in enumerate(traced):
        xa = torch.from_numpy(feats[b:b + 1])                # the decoder alone: both sides attend the same features
        infos.append(check_clip(model, ora, st, oo, res[b], slot, xa, dtype_name, budget))
        assert infos[-1]["n_pos"] >= len(res[b].tokens)
        ref = OW.greedy_decode(ora, st, xa, OW.DecodingOptions(max_new_tokens=64, **kw))
        k = first_fork(res[b].tokens[:64], ref.tokens)
        if k is not None:                                    # (this mel is not the one the seed was picked on: a near-tie may occur)
            assert_fork_explained(infos[-1], ref, k)
    base_logits = [i["hip"] for i in infos]
    for flags in (1, 2, 3):
        model.set_debug(flags)
        again = HW.GreedyDecoder(model, HW.DecodingOptions(**kw)).decode(mel)
        for b in range(B):
            assert again[b].tokens == res[b].tokens, (flags, b)
            assert np.array_equal(np.float32(again[b].avg_logprob), np.float32(res[b].avg_logprob), equal_nan=True) or \
                abs(again[b].avg_logprob - res[b].avg_logprob) <= 1e-5          # flag 2: the one-workgroup head sums in another order
        for slot in range(len(traced)):
            np.testing.assert_array_equal(model.read_logit_trace(slot, 0, infos[slot]["n_pos"]), base_logits[slot])
    model.set_debug(0)
    model.close()


def test_all_masked_step_quirk(ctx):
    """Kept on purpose: the Swift port evaluates the timestamp heuristic on RAW logits (WhisperDecoding.swift:299-322), so right after
    the rule-forced first timestamp it can mask every token; MLX then yields token 0 and a NaN log-prob.  A 'lecun' checkpoint
    (flat softmax) triggers it at step 1 in every clip; HIP must mirror token 0 and the NaN average."""
    from mlx_swift_audio_amd import whisper as HW
    dims, ora, model = _setup(ctx, "micro.en", "f16", 5, style="lecun")
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    sup = OW.synthetic_suppress_list(st)
    mel = _mel(dims, 2, 1, "f16")
    kw = dict(timestamps=True, suppress_ids=sup, blank_ids=[220, 50255], max_new_tokens=6)
    oo = OW.DecodingOptions(**kw)
    model.trace_logits([0, 1])
    res = HW.GreedyDecoder(model, HW.DecodingOptions(**kw)).decode(mel)
    xa = ora.encode(mel)
    for b in range(2):
        ref = OW.greedy_decode(ora, st, xa[b:b + 1], oo)
        assert ref.tokens[1] == 0 and np.isnan(ref.avg_logprob)                 # the quirk is present in the oracle's run
        assert res[b].tokens[:2] == ref.tokens[:2] and np.isnan(res[b].avg_logprob)
        check_clip(model, ora, st, oo, res[b], b, xa[b:b + 1], "f16", 6)
    model.close()


def test_trace_hook_errors(ctx):
    import mlx_swift_audio_amd as m
    dims, ora, model = _setup(ctx, "micro.en", "f16", 77)
    with pytest.raises(m.MiaError):
        model.read_logit_trace(0, 0, 1)                     # no trace requested
    with pytest.raises(m.MiaError):
        model.set_debug(4)
    model.trace_logits([3])
    model.encode(_mel(dims, 2, 0, "f16"))
    from mlx_swift_audio_amd import whisper as HW
    with pytest.raises(m.MiaError):                         # traced row 3 is not in a batch of 2
        model.decode_greedy(HW.DecodingOptions(max_new_tokens=2))
    model.trace_logits([])
    model.decode_greedy(HW.DecodingOptions(max_new_tokens=2))
    model.close()
