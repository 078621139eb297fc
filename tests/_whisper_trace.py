"""Shared checker of the Whisper step path (test infrastructure): the step graph's own logits, traced through the test-only ABI hook
mia_whisper_trace_logits, against the oracle at EVERY position, plus an exact replay of the decode head's decisions.

What it proves for one clip of a finished mia_whisper_decode_greedy call (temperature 0):

1. logits: row p of the trace (what the captured step graph computed after consuming tokens[0..p], self-KV rows 0..p, positional row p)
   equals the oracle's teacher-forced logits on the SAME token prefix (oracle/whisper.py:teacher_forced_logits, TextDecoder.swift:53-96)
   within the written 16-bit tolerance -- at every position up to the last one decoded, so it does not stop at a fork: a broken
   self-attention cache, a wrong positional row, audio-independent cross-attention or a bad graph replay shows up as an O(1) error.
2. head: re-running the oracle's rule code (filter_logits: suppress / timestamp rules / raw-logit heuristic, argmax, log-prob sum;
   WhisperDecoding.swift:186-350) on the TRACED logits along the emitted sequence reproduces every emitted id exactly and the
   reported avg_logprob to fp32 rounding -- the head kernels implement the rules bit for bit on their own inputs.
Together: the ids are an exact function of logits that are within tolerance of the oracle's.  Where the oracle's free run picks another
token than HIP at some step, (1) + (2) imply its top-2 margin there is at most the sum of the two measured logit errors."""
import numpy as np
import torch

from oracle import whisper as OW

# tolerances of the traced logits against the oracle's, per storage type:
#   rms over the vocabulary of |delta| <= TOL_RMS * std(oracle row);   max |delta| <= TOL_MAX * max |oracle row|
TOL_RMS = {"f16": 0.004, "bf16": 0.03}
TOL_MAX = {"f16": 0.012, "bf16": 0.08}


def check_clip(model, ora, st, oo, result, slot, xa, dtype_name, budget, tol_scale=1.0):
    """result: the clip's DecodingResult from HIP; slot: its trace slot; xa: torch [1, T, D] features fed to the oracle decoder.
    Returns a dict of measured quantities (noise_rms in logit units, per-step arrays)."""
    init, _ = OW.initial_tokens(st, oo)
    toks = list(init) + list(result.tokens)
    if len(result.tokens) < budget:                     # the run ended on EOT (stripped from the output)
        toks.append(st.eot)
    n = len(toks)
    hip = model.read_logit_trace(slot, 0, n - 1)        # row p predicts toks[p + 1]
    assert np.isfinite(hip).all(), "a traced row was never written (NaN fill pattern)"
    ref = OW.teacher_forced_logits(ora, xa, toks[:n - 1])
    d = np.abs(hip - ref)
    rms = np.sqrt((d.astype(np.float64) ** 2).mean(axis=1))
    rel_rms = rms / ref.std(axis=1)
    rel_max = d.max(axis=1) / np.abs(ref).max(axis=1)
    worst = int(rel_rms.argmax())
    assert rel_rms.max() <= TOL_RMS[dtype_name] * tol_scale, (dtype_name, "rms", worst, float(rel_rms.max()))
    assert rel_max.max() <= TOL_MAX[dtype_name] * tol_scale, (dtype_name, "max", int(rel_max.argmax()), float(rel_max.max()))
    ids, margins, dists, avg = OW.replay_rules(hip, toks, len(init), st, oo)
    emitted = toks[len(init):]
    for k, (a, b) in enumerate(zip(ids, emitted)):
        if a != b:
            # the only legal disagreement: the heuristic's threshold compare (ts_logprob > max_text_logprob) decided within fp32
            # rounding of equality (device __expf / __logf against torch's) -- then either branch is the rule's outcome
            assert dists[k] is not None and abs(dists[k]) < 1e-4, (k, a, b, margins[k], dists[k])
    if np.isnan(avg) or np.isnan(result.avg_logprob):
        assert np.isnan(avg) and np.isnan(result.avg_logprob), (avg, result.avg_logprob)
    else:
        assert abs(avg - result.avg_logprob) <= 2e-4 * max(1.0, abs(avg)), (avg, result.avg_logprob)
    return {"n_pos": n - 1, "noise_rms": float(np.sqrt((d.astype(np.float64) ** 2).mean())), "rel_rms_max": float(rel_rms.max()),
            "rel_max_max": float(rel_max.max()), "d": d, "ref": ref, "hip": hip, "tokens": toks, "n_init": len(init), "margins_hip": margins}


def first_fork(a, b):
    return next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None if len(a) == len(b) else min(len(a), len(b)))


def assert_fork_explained(info, ref_result, k):
    """HIP and the oracle's free run split at generated index k (same prefix, same features, so the oracle's logits there are the
    teacher-forced row).  With check_clip's two properties there are exactly two ways this can happen, both asserted numerically:
      * same masks, other argmax: the oracle's top-2 margin is at most the sum of the measured logit errors of the two tokens;
      * the raw-logit timestamp heuristic (ts_logprob > max_text_logprob, WhisperDecoding.swift:299-322) decided the other way:
        logsumexp and max are 1-Lipschitz in the sup norm, so its distance to the threshold in the oracle is at most
        2 x the largest logit error of that row."""
    p = info["n_init"] - 1 + k
    a = info["tokens"][info["n_init"] + k]
    c = ref_result.tokens[k]
    bound = float(info["d"][p, a] + info["d"][p, c])
    hd = ref_result.heur[k] if ref_result.heur else None
    flip_ok = hd is not None and abs(hd) <= 2.0 * float(info["d"][p].max())
    assert ref_result.margins[k] <= bound + 1e-6 or flip_ok, (k, ref_result.margins[k], bound, hd, float(info["d"][p].max()))
    return bound


def explain_fork_other_features(info, ora, xa_other, ref_result, k):
    """The same statement for a free run of the oracle on OTHER features than check_clip used (its own fp32 encoder output: the error
    then includes the 16-bit encoder).  Returns (oracle margin, |d| of HIP's token, |d| of the oracle's token, logit std) at the fork."""
    p = info["n_init"] - 1 + k
    row = OW.teacher_forced_logits(ora, xa_other, info["tokens"][:p + 1])[p]
    d = np.abs(info["hip"][p] - row)
    a, c = info["tokens"][info["n_init"] + k], ref_result.tokens[k]
    hd = ref_result.heur[k] if ref_result.heur else None
    flip_ok = hd is not None and abs(hd) <= 2.0 * float(d.max())
    assert ref_result.margins[k] <= float(d[a] + d[c]) + 1e-6 or flip_ok, (k, ref_result.margins[k], float(d[a]), float(d[c]), hd)
    return ref_result.margins[k], float(d[a]), float(d[c]), float(row.std())


def nondegenerate(ref_results, n_new, min_distinct=16):
    """The ORACLE's own free runs: varied ids, finite log-probs, clips differ.  Returns the smallest top-2 margin."""
    for r in ref_results:
        assert len(r.tokens) >= min(n_new, 32), len(r.tokens)
        assert len(set(r.tokens)) >= min_distinct, ("degenerate sequence", len(set(r.tokens)), r.tokens)
        assert np.isfinite(r.avg_logprob), r.avg_logprob
    for i in range(len(ref_results)):
        for j in range(i + 1, len(ref_results)):
            assert ref_results[i].tokens != ref_results[j].tokens, ("clips decode identically", i, j)
    return min(min(r.margins) for r in ref_results)
