"""Measure the 16-bit noise of the Whisper step path: HIP step-graph logits (mia_whisper_trace_logits) against the oracle's
teacher-forced logits on the tokens HIP emitted, per position.  Prints max / rms |delta| and the logit spread.
usage: python tools/logit_noise.py <dims> <bf16|f16> <style> <seed> <B> <n_new> [timestamps 0|1]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mlx_swift_audio_amd as m
from mlx_swift_audio_amd import synthetic as S
from mlx_swift_audio_amd import whisper as HW
from oracle import whisper as OW


def main():
    dims_name, rt, style, seed, B, n_new = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    ts = len(sys.argv) < 8 or sys.argv[7] == "1"
    dims = S.DIMS[dims_name]
    ctx = m.Context(0)
    w = S.synthetic_weights(dims, seed=seed, style=style, round_to=rt)
    model = HW.WhisperModel.load(ctx, dims, w, m.BF16 if rt == "bf16" else m.F16)
    ora = OW.WhisperOracle(dims, w)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    rng = np.random.default_rng(1)
    mel = S.round_array((0.5 * rng.standard_normal((B, 2 * dims.n_audio_ctx, dims.n_mels))).astype(np.float32), rt)
    sup = S.synthetic_suppress_list(st)
    o = HW.DecodingOptions(timestamps=ts, suppress_ids=sup, blank_ids=[220, 50255], max_new_tokens=n_new)
    oo = OW.DecodingOptions(timestamps=ts, suppress_ids=sup, blank_ids=[220, 50255], max_new_tokens=n_new)
    clips = list(range(min(B, 4)))
    model.trace_logits(clips)
    t0 = time.time()
    res = HW.GreedyDecoder(model, o).decode(mel)
    feats = model.audio_features()
    xa_o = ora.encode(mel)
    import torch
    init, _ = OW.initial_tokens(st, oo)
    for slot, b in enumerate(clips):
        toks = init + res[b].tokens
        n = len(toks)
        hip = model.read_logit_trace(slot, 0, n - 1)
        for name, xa in (("hip-feats", torch.from_numpy(feats[b:b + 1])), ("ora-feats", xa_o[b:b + 1])):
            ref = OW.teacher_forced_logits(ora, xa, toks[:n - 1])
            d = np.abs(hip - ref)
            print(f"clip {b} {name}: n {n} max|d| {d.max():.5f} rms {np.sqrt((d ** 2).mean()):.6f} logit std {ref.std():.4f} "
                  f"worst pos {int(d.max(axis=1).argmax())} rel {d.max() / ref.std():.5f}", flush=True)
        ids, margins, dists, avg = OW.replay_rules(hip, toks, len(init), st, oo)
        ok = ids == res[b].tokens
        r = OW.greedy_decode(ora, st, xa_o[b:b + 1], oo)
        k = next((i for i, (a, c) in enumerate(zip(res[b].tokens, r.tokens)) if a != c), None)
        print(f"  replay==hip {ok}  avg hip {res[b].avg_logprob:.5f} replay {avg:.5f} oracle {r.avg_logprob:.5f}  oracle min margin {min(r.margins):.4f} "
              f"first fork vs oracle {k} distinct {len(set(r.tokens))}", flush=True)
    print("time", time.time() - t0)


if __name__ == "__main__":
    main()
