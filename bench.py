#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native speech-inference hot path.

Metric (BASELINE.json): audio-seconds transcribed per wall-second, Whisper large-v3-turbo, bf16, 32 x 30 s synthetic
clips per GPU (weak scaling: every rank transcribes its own 32 clips, then ONE RCCL all-gather of the token ids).
A "step" = one pass of the hot path over the batch: log-mel -> encoder -> cross-KV -> greedy decode (one 30 s window per
clip, T=0, timestamps on, max_tokens 448) with pcm already resident in HBM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel class, measured
live with HIP events on the library's stream) and `cpu_baseline` (the fp32 CPU restatement, oracle/, timed on the
host cores of this box on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# The replicas' HIP streams must land on distinct hardware queues: with ROCm's default of 4 queues per process two of three replica
# streams share one and their kernels serialise (three concurrent decoders: 464 ms instead of 377 ms per round, measured).  The knob
# is read when the runtime initialises, i.e. before the first `import torch`; the launcher's children inherit it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def cpu_baseline(dims_name: str, seed: int, n_threads: int, max_new_tokens: int, repeats: int = 3):
    """fp32 CPU restatement of the same pipeline on ONE of the clips (bounded sample), all stages timed; `repeats` complete runs,
    the MEDIAN is reported (one sample right after GPU work swung 2.1 -> 1.7 audio-s/s between rounds) and every run is listed."""
    import torch
    from oracle import logmel as OL
    from oracle import whisper as OW
    torch.set_num_threads(n_threads)
    dims = OW.DIMS[dims_name]
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    weights = OW.synthetic_weights(dims, seed=seed, style="survey", round_to="bf16")
    model = OW.WhisperOracle(dims, weights)
    clip = OL.synth_clip(0)
    o = OW.DecodingOptions(suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220], max_new_tokens=max_new_tokens)
    runs = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        mel = OL.whisper_log_mel_spectrogram(clip, dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES]
        mel = OW.round_array(mel, "bf16")[None]
        t1 = time.perf_counter()
        xa = model.encode(mel)
        t2 = time.perf_counter()
        r = OW.greedy_decode(model, st, xa, o)
        t3 = time.perf_counter()
        n_tok = max(len(r.margins), 1)
        full_tokens = dims.n_text_ctx - len(r.initial_tokens)
        # decode is linear in the number of steps: scale the measured steps to the full 445-step budget the GPU run executes
        t_dec_full = (t3 - t2) * full_tokens / n_tok
        runs.append((30.0 / ((t1 - t0) + (t2 - t1) + t_dec_full), t1 - t0, t2 - t1, t3 - t2, n_tok, full_tokens))
    runs.sort()
    v, a, b, c, n_tok, full_tokens = runs[len(runs) // 2]
    return {"value": round(v, 3), "unit": "audio-sec/s", "cores": n_threads, "kind": "port",
            "sample": (f"1 of the 30 s clips, fp32 torch-CPU restatement (oracle/), median of {len(runs)} complete runs "
                       f"({', '.join('%.2f' % r[0] for r in runs)} audio-sec/s): log-mel {a:.2f}s + encoder {b:.2f}s + "
                       f"{n_tok} greedy steps {c:.2f}s scaled to {full_tokens} steps")}


def visible_gpu_count() -> int:
    """GPUs this process may use, WITHOUT touching the HIP runtime (the launcher must stay HIP-free: it spawns the ranks): the KFD
    topology lists one node per agent, GPUs are the nodes with simd_count > 0; ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES narrow the set like they do for the runtime."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n: int) -> int:
    """Start n rank processes of this script on one node (plain child processes, subprocess.Popen -- not torch.distributed.run) and
    wait for them; rank 0's JSON line goes to our stdout.  This launcher never initialises HIP: devices are counted from sysfs."""
    import socket
    import subprocess
    dry = "--dry-run" in sys.argv
    rehearsal = os.environ.get("MIA_BENCH_REHEARSAL") == "1"
    if not rehearsal and not dry:
        have = visible_gpu_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    for r, c in enumerate(rcs):
        if c != 0:
            print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
    return max(abs(c) for c in rcs)


class Exchanger:
    """The ONE exchange channel of a rank (one all-gather of token ids per pass).  With R replicas decoding on R streams
    from R host threads, letting every replica issue its own collective would make the order of collectives timing-dependent and
    different from rank to rank -- the documented deadlock shape of NCCL-family libraries.  Here every rank deals pass i to replica
    i % R, ONE thread per rank issues the gathers strictly in pass order 0, 1, 2, ... on ONE communicator / stream, and a gather
    waits (stream-side) on the event its pass recorded behind the decode.  Same order on every rank by construction."""

    def __init__(self, issue):
        import threading
        self.issue = issue                   # issue(payload): enqueue ONE gather on the exchange stream (never blocks on the GPU)
        self.cv = threading.Condition()
        self.pending = {}
        self.next = 0
        self.stop_at = None
        self.error = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        self.order = []                      # pass ids in the order issued (asserted == 0, 1, 2, ... by the tests)

    def submit(self, pass_id, payload):
        with self.cv:
            self.pending[pass_id] = payload
            self.cv.notify_all()

    def _run(self):
        while True:
            with self.cv:
                while self.next not in self.pending and (self.stop_at is None or self.next < self.stop_at):
                    self.cv.wait()
                if self.next not in self.pending:
                    return
                payload = self.pending.pop(self.next)
            try:
                self.issue(payload)
            except BaseException as e:       # surface in close(); keep consuming so that no submitter waits forever
                self.error = self.error or e
            self.order.append(self.next)
            with self.cv:
                self.next += 1
                self.cv.notify_all()

    def wait_issued(self, pass_id):
        """Block (host side) until the gather of `pass_id` has been enqueued."""
        with self.cv:
            while self.next <= pass_id:
                self.cv.wait()

    def close(self, n_passes):
        with self.cv:
            self.stop_at = n_passes
            self.cv.notify_all()
        self.thread.join()
        if self.error:
            raise self.error


def dry_run(args, rank, world):
    """Control flow of an N-rank run WITHOUT a GPU (tests/test_multi_rank.py): R replica threads per rank 'decode' passes (they write
    rank- and pass-stamped token rows after a jittered sleep, so ranks and replicas finish in different orders), the Exchanger issues
    one gloo all-gather per pass in pass order, every rank checks every gathered row.  Prints the contract's JSON line with
    data = 'dry-run'."""
    import random
    import threading
    import torch
    import torch.distributed as dist
    from mlx_swift_audio_amd import parallel as P
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, L, R = args.batch, 448, max(1, min(args.replicas, args.steps))
    total = args.warmup * R + args.steps
    got = {}

    def issue(payload):
        pid, toks, cnt = payload
        got[pid] = P.gather_tokens(toks, cnt, world, max_shard=B) if world > 1 else (toks, cnt)

    ex = Exchanger(issue)
    rnd = random.Random(1000 * rank + 7)

    def replica(r):
        for pid in range(r, total, R):
            time.sleep(rnd.random() * 0.02)
            toks = torch.full((B, L), 1000 * pid + rank, dtype=torch.int32)
            toks[:, 0] = torch.arange(B, dtype=torch.int32)
            ex.submit(pid, (pid, toks, torch.full((B,), pid % 7, dtype=torch.int32)))
            ex.wait_issued(pid)                       # the replica's buffers are free again once its gather has been enqueued

    t0 = time.perf_counter()
    ths = [threading.Thread(target=replica, args=(r,)) for r in range(R)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    ex.close(total)
    elapsed = time.perf_counter() - t0
    assert ex.order == list(range(total)), ex.order
    for pid in range(total):
        toks, cnt = got[pid]
        assert toks.shape == (B * world, L) and cnt.shape == (B * world,)
        for rk in range(world):
            blk = toks[rk * B:(rk + 1) * B]
            assert bool((blk[:, 1:] == 1000 * pid + rk).all()) and bool((blk[:, 0] == torch.arange(B, dtype=torch.int32)).all()), (pid, rk)
        assert bool((cnt == pid % 7).all())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "audio-sec/s (Whisper large-v3-turbo b=32) at 1/2/4/8 GPU; codec samples/s", "value": 0.0, "unit": "audio-sec/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(total, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "dry-run (no GPU work: launcher + exchange order only)",
                          "config": {"workload": "dry-run", "replicas": R, "passes_exchanged": total, "exchange_order_ok": True}}), flush=True)


def config0_leg(ctx, torch, dtype_name: str = "f16"):
    """BASELINE.json configs[0]: Whisper tiny.en, greedy transcribe of ONE 10 s mono clip (the reference's CPU-runnable plumbing case).
    GPU: the same C-ABI path (log-mel + encode + 445-step decode, f16 like the reference's storage); CPU: the oracle at all cores
    and at 1 thread, complete run, nothing extrapolated."""
    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import synthetic as S
    from mlx_swift_audio_amd import whisper as HW
    import torch as _t
    from oracle import logmel as OL
    from oracle import whisper as OW
    dims = S.DIMS["tiny.en"]
    weights = S.synthetic_weights(dims, seed=0, style="survey", round_to=dtype_name)
    model = HW.WhisperModel.load(ctx, dims, weights, m.F16 if dtype_name == "f16" else m.BF16)
    o = HW.DecodingOptions(suppress_ids=S.synthetic_suppress_list(model.special), blank_ids=[220])
    clip = S.synth_clip(0)[:160000]
    model.transcribe_windows([clip], o)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        res = model.transcribe_windows([clip], o)[0]
    gpu_s = (time.perf_counter() - t0) / reps
    model.close()
    out = {"workload": "Whisper tiny.en f16, one 10 s clip, log-mel + encode + greedy decode (445 generated tokens: random-init weights never emit EOT)",
           "gpu": {"audio_s_per_s": round(10.0 / gpu_s, 1), "ms": round(gpu_s * 1e3, 2), "generated_tokens": len(res.tokens), "note": "host-inclusive wall clock (pcm and tokens cross PCIe)"}}
    ora = OW.WhisperOracle(dims, weights)
    st = OW.SpecialTokens.for_vocab(dims.n_vocab)
    oo = OW.DecodingOptions(suppress_ids=OW.synthetic_suppress_list(st), blank_ids=[220])
    ncpu = min(16, len(os.sched_getaffinity(0)))
    for nt in (ncpu, 1):
        _t.set_num_threads(nt)
        t0 = time.perf_counter()
        mel = OW.round_array(OL.whisper_log_mel_spectrogram(clip, dims.n_mels, padding=OL.N_SAMPLES)[:OL.N_FRAMES], dtype_name)[None]
        r = OW.greedy_decode(ora, st, ora.encode(mel), oo)
        dt = time.perf_counter() - t0
        out[f"cpu_{nt}_threads" if nt != 1 else "cpu_1_thread"] = {"audio_s_per_s": round(10.0 / dt, 3), "s": round(dt, 2), "generated_tokens": len(r.tokens), "kind": "port"}
    _t.set_num_threads(ncpu)
    return out


def lm_bench(ctx, torch, name: str, batch: int):
    """BASELINE.json configs[2] ("Orpheus-3B TTS: Llama-3 backbone autoregress + SNAC codec decode"), the LM half, and configs[3]'s
    Qwen2LM: random-init weights, 64-token prompts, 210 sampled tokens (30 SNAC frames); one sequence on bf16 weights and on packed
    MLX-affine 4-bit weights (the reference's default checkpoint format, OrpheusWeightLoader.swift:28-60; for the TIMING leg the packed
    codes / scales are random words of the right shape -- quantising 3.3 G weights in numpy would take minutes and changes no byte
    count; the bit-identity of the packed step with the expanded checkpoint is tests/test_lm_gpu.py's job), then `batch` sentences side
    by side.  ms_per_token excludes the prompt pass; frac = algorithmic weight bytes per token / time / 8 TB/s."""
    from mlx_swift_audio_amd import lm as HL
    from mlx_swift_audio_amd import synthetic as S
    import mlx_swift_audio_amd as m
    cfg = S.LM_CONFIGS[name]
    t0 = time.time()
    w = S.lm_weights(cfg, seed=0, dtype=np.float16)
    model = HL.CausalLM.load(ctx, cfg, w, m.BF16)
    rng = np.random.default_rng(0)
    lin = (["model.embed_tokens"] if cfg.tie_embeddings else ["lm_head"])
    for l in range(cfg.n_layers):
        lin += [f"model.layers.{l}.self_attn.{n}_proj" for n in "qkvo"] + [f"model.layers.{l}.mlp.{n}_proj" for n in ("gate", "up", "down")]
    packed = {}
    for n in lin:
        N, K = w[n + ".weight"].shape
        packed[n + ".weight"] = rng.integers(0, 1 << 32, (N, K // 8), dtype=np.uint32)
        packed[n + ".scales"] = np.full((N, K // 64), 0.004, np.float16)
        packed[n + ".biases"] = np.full((N, K // 64), -0.03, np.float16)
    del w
    model.attach_q4(packed)
    del packed
    log(f"[bench] {name} checkpoint (bf16 + packed q4) ready in {time.time() - t0:.1f}s")
    n_new, n_prompt = 210, 64
    stop = (cfg.vocab - 1,)
    prompt = rng.integers(0, min(128000, cfg.vocab), n_prompt).tolist()
    u = rng.random(n_new).astype(np.float32)
    params = cfg.vocab * cfg.hidden + cfg.n_layers * ((cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim * cfg.hidden + cfg.hidden * cfg.n_heads * cfg.head_dim + 3 * cfg.inter * cfg.hidden)

    def one(q4: bool):
        model.use_q4(q4)
        model.generate(prompt, u, max_new_tokens=16, stop_ids=stop)          # graph capture for this weight format
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(prompt, u, max_new_tokens=1, stop_ids=stop)           # batched prompt pass + the first step
        d_prompt = time.perf_counter() - t0
        t0 = time.perf_counter()
        gen = model.generate(prompt, u, max_new_tokens=n_new, stop_ids=stop)
        dt = time.perf_counter() - t0
        steps = len(gen) - 1
        bytes_tok = params * (0.5 + 4.0 / 64) if q4 else 2.0 * params        # q4: 4 bits + (scale, bias) 2 x 16 bit per 64 weights
        ms = (dt - d_prompt) / max(steps, 1) * 1e3
        r = {"weights": "mlx-affine q4 g64, packed step" if q4 else "bf16", "generated_tokens": len(gen), "prompt_pass_plus_first_step_ms": round(d_prompt * 1e3, 2),
             "ms_per_token": round(ms, 4), "tokens_per_s": round(1e3 / ms, 1), "weight_GB_per_token": round(bytes_tok / 1e9, 3),
             "GBs_algorithmic": round(bytes_tok / (ms * 1e-3) / 1e9, 1), "frac": round(bytes_tok / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if batch > 1:
            model.set_batch(batch)
            prompts = [rng.integers(0, min(128000, cfg.vocab), n_prompt).tolist() for _ in range(batch)]
            ub = rng.random((batch, n_new)).astype(np.float32)
            model.generate_batch(prompts, ub, max_new_tokens=16, stop_ids=stop)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = model.generate_batch(prompts, ub, max_new_tokens=n_new, stop_ids=stop)
            db = time.perf_counter() - t0
            ntok = sum(len(o) for o in outs)
            r["batched"] = {"sequences": batch, "tokens_per_s": round(ntok / db, 1), "audio_seconds_per_second": round(ntok / 7 * 2048 / 24000.0 / db, 1)}
            model.set_batch(1)
        return r

    res = {"model": name, "prompt_tokens": n_prompt, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "bf16": one(False), "q4": one(True)}
    model.close()
    # ---- CosyVoice2's Qwen2LM (configs[3]): Qwen2-0.5B backbone, 300 prompt rows -> 300 speech tokens (12 s), RAS sampler
    try:
        qcfg = S.LM_CONFIGS["qwen2-0.5b"]
        qw = S.lm_weights(qcfg, seed=0, dtype=np.float16)
        qw.update(S.qwen2lm_extra_weights(qcfg, 6561, seed=0))
        qm = HL.CausalLM.load(ctx, qcfg, qw, m.BF16)
        del qw
        n_rows, n_gen = 300, 300
        x = rng.standard_normal((n_rows, qcfg.hidden)).astype(np.float32)
        uq = rng.random(4 * n_gen + 64).astype(np.float32)

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = qm.generate_ras(x, uq, n, n, 6561)
            return time.perf_counter() - t0, out
        run(8)
        d1, _ = run(1)
        dt, out = run(n_gen)
        qparams = 6564 * qcfg.hidden + qcfg.n_layers * ((qcfg.n_heads + 2 * qcfg.n_kv_heads) * qcfg.head_dim * qcfg.hidden + qcfg.hidden * qcfg.n_heads * qcfg.head_dim + 3 * qcfg.inter * qcfg.hidden)
        ms = (dt - d1) / max(len(out) - 1, 1) * 1e3
        res["qwen2lm"] = {"model": "qwen2-0.5b backbone + llm_decoder head (Qwen2LM.inference)", "weights": "bf16", "prompt_rows": n_rows, "generated_tokens": len(out),
                          "prompt_pass_plus_first_step_ms": round(d1 * 1e3, 2), "ms_per_token": round(ms, 4), "weight_GB_per_token": round(2.0 * qparams / 1e9, 3),
                          "GBs_algorithmic": round(2.0 * qparams / (ms * 1e-3) / 1e9, 1), "frac": round(2.0 * qparams / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "audio_seconds_per_second": round(len(out) / 25.0 / dt, 1)}
        qm.close()
    except Exception as e:
        res["qwen2lm"] = {"error": repr(e)}
    return res


def codec_bench(ctx, torch):
    """Second half of BASELINE.json's metric ("codec samples/s"): SNAC 24 kHz decode of one Orpheus chunk (1200 tokens ->
    171 frames -> 350 208 samples) and DAC speech decode of 10 s (750 code steps), random-init weights,
    codes/noise resident on the host side of the ABI excluded: timed with device-resident inputs via HIP events."""
    import ctypes as C
    from mlx_swift_audio_amd import codec as HC
    from mlx_swift_audio_amd import synthetic as S
    res = {}
    rng = np.random.default_rng(7)
    ctx.lib.mia_profile_codec_bytes.restype = C.c_double
    ctx.lib.mia_profile_codec_bytes.argtypes = [C.c_int]

    def alg(run, ms):
        """algorithmic bytes of one call (each fused op's input rows once + outputs once + weights once, counted by the
        library's launchers: mia_profile_codec_bytes) over the measured time, against the 8 TB/s HBM peak."""
        ctx.lib.mia_profile_codec_bytes(1)
        run()
        b = float(ctx.lib.mia_profile_codec_bytes(1))
        torch.cuda.synchronize()
        gbs = b / (ms * 1e-3) / 1e9
        return {"algorithmic_MB": round(b / 1e6, 1), "GBs_algorithmic": round(gbs, 1), "bound": "hbm", "peak": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4)}

    # ---- SNAC
    cfg = S.SNAC_CONFIGS["snac_24khz"]
    dec = HC.SNACDecoder.load(ctx, cfg, S.snac_weights(cfg, 0))
    N = 171
    codes = [torch.from_numpy(rng.integers(0, cfg.codebook_size, N * (4 // s)).astype(np.int32)).cuda() for s in cfg.vq_strides]
    T0 = N * 4
    n_noise, n_out = dec.noise_len(T0), dec.output_len(T0)
    noise = torch.randn(n_noise, device="cuda")
    pcm = torch.empty(n_out, device="cuda")
    ptrs = (C.c_void_p * 3)(*[c.data_ptr() for c in codes])
    n_codes = np.asarray([c.numel() for c in codes], np.int32)
    ns = C.c_int64(0)

    def run_snac():
        ctx.check(ctx.lib.mia_snac_decode(dec.h, ptrs, n_codes.ctypes.data, 3, noise.data_ptr(), n_noise, pcm.data_ptr(), n_out, C.byref(ns), 1))

    for _ in range(2):
        run_snac()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        run_snac()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res["snac_24khz_decode"] = {"samples_per_s": round(n_out / (ms * 1e-3), 0), "ms": round(ms, 3), "samples": n_out,
                                "realtime_factor": round(n_out / 24000.0 / (ms * 1e-3), 1), **alg(run_snac, ms)}
    dec.close()
    # ---- DAC
    dcfg = S.DAC_CONFIGS["dac_speech"]
    dd = HC.DACCodec.load(ctx, dcfg, S.dac_weights(dcfg, 0))
    T = 750
    dcodes = torch.from_numpy(rng.integers(0, dcfg.codebook_size, (dcfg.n_codebooks, T)).astype(np.int32)).cuda()
    d_out = dd.output_len(T)
    dpcm = torch.empty(d_out, device="cuda")

    def run_dac():
        ctx.check(ctx.lib.mia_dac_decode(dd.h, dcodes.data_ptr(), dcfg.n_codebooks, T, dpcm.data_ptr(), d_out, C.byref(ns), 1))

    for _ in range(2):
        run_dac()
    e0.record()
    for _ in range(reps):
        run_dac()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res["dac_speech_decode"] = {"samples_per_s": round(d_out / (ms * 1e-3), 0), "ms": round(ms, 3), "samples": d_out,
                                "realtime_factor": round(d_out / 24000.0 / (ms * 1e-3), 1), **alg(run_dac, ms)}
    dd.close()
    # ---- HiFT (CosyVoice2 vocoder): 10 s of 50 Hz mel -> 240 000 samples, source noise included as an input
    from mlx_swift_audio_amd import hift as HH
    hcfg = S.HIFT_CONFIGS["hift_cosyvoice2"]
    hg = HH.HiFTGenerator.load(ctx, hcfg, S.hift_weights(hcfg, 0))
    T = 500
    h_out = T * hg.up
    mel = (torch.randn(hcfg.in_channels, T, device="cuda") * 1.5 - 2.0).contiguous()
    hnoise = torch.randn(h_out, hcfg.nb_harmonics + 1, device="cuda")
    hpcm = torch.empty(h_out, device="cuda")

    def run_hift():
        ctx.check(ctx.lib.mia_hift_vocode(hg.h, mel.data_ptr(), T, hnoise.data_ptr(), None, 0, hpcm.data_ptr(), None, 1))

    for _ in range(2):
        run_hift()
    e0.record()
    for _ in range(reps):
        run_hift()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res["hift_cosyvoice2_vocode"] = {"samples_per_s": round(h_out / (ms * 1e-3), 0), "ms": round(ms, 3), "samples": h_out,
                                     "realtime_factor": round(h_out / 24000.0 / (ms * 1e-3), 1), **alg(run_hift, ms)}
    # the same utterance x 8 in one stacked call (mia_hift_vocode_batch: every convolution once over 8 sequences)
    NB = 8
    mel8 = mel.unsqueeze(0).repeat(NB, 1, 1).contiguous()
    noise8 = hnoise.unsqueeze(0).repeat(NB, 1, 1).contiguous()
    pcm8 = torch.empty(NB * h_out, device="cuda")
    T8 = np.full(NB, T, np.int32)

    def run_hift8():
        ctx.check(ctx.lib.mia_hift_vocode_batch(hg.h, mel8.data_ptr(), T8.ctypes.data, NB, noise8.data_ptr(), pcm8.data_ptr(), 1))

    for _ in range(2):
        run_hift8()
    e0.record()
    for _ in range(reps):
        run_hift8()
    e1.record()
    torch.cuda.synchronize()
    ms8 = e0.elapsed_time(e1) / reps
    same = bool(torch.equal(pcm8.view(NB, h_out)[NB - 1], hpcm)) and bool(torch.equal(pcm8.view(NB, h_out)[0], hpcm))
    res["hift_cosyvoice2_vocode_batch8"] = {"samples_per_s": round(NB * h_out / (ms8 * 1e-3), 0), "ms": round(ms8, 3), "utterances": NB,
                                            "ms_per_utterance": round(ms8 / NB, 3), "realtime_factor": round(NB * h_out / 24000.0 / (ms8 * 1e-3), 1),
                                            "equals_single_call": same}
    hg.close()
    # ---- CosyVoice2 flow: 375 new + 150 prompt speech tokens (15 s + 6 s) -> 750 new mel frames, 10 Euler steps with CFG
    from mlx_swift_audio_amd import flow as HFL
    fcfg = S.FLOW_CONFIGS["flow_cosyvoice2"]
    fm = HFL.FlowModule.load(ctx, fcfg, S.flow_weights(fcfg, 0))
    n_tok, n_prompt = 375, 150
    Tm = 2 * (n_tok + n_prompt)
    tok = torch.from_numpy(rng.integers(0, fcfg.vocab_size, n_tok).astype(np.int32)).cuda()
    ptok = torch.from_numpy(rng.integers(0, fcfg.vocab_size, n_prompt).astype(np.int32)).cuda()
    pfeat = torch.randn(2 * n_prompt, 80, device="cuda")
    spk = torch.randn(fcfg.spk_embed_dim, device="cuda")
    zz = torch.randn(80, Tm, device="cuda")
    fmel = torch.empty(80, Tm - 2 * n_prompt, device="cuda")

    def run_flow():
        ctx.check(ctx.lib.mia_flow_inference(fm.h, tok.data_ptr(), n_tok, ptok.data_ptr(), n_prompt, pfeat.data_ptr(), 2 * n_prompt,
                                             spk.data_ptr(), zz.data_ptr(), 0, fmel.data_ptr(), 1))

    run_flow()
    e0.record()
    freps = 3
    for _ in range(freps):
        run_flow()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / freps
    n_new = Tm - 2 * n_prompt
    res["flow_cosyvoice2_inference"] = {"mel_frames_per_s": round(n_new / (ms * 1e-3), 0), "ms": round(ms, 3), "new_mel_frames": n_new,
                                        "total_frames": Tm, "euler_steps": fcfg.n_timesteps,
                                        "realtime_factor": round(n_new / 50.0 / (ms * 1e-3), 1), **alg(run_flow, ms)}
    # the same utterance 8 times through ONE pass (mia_flow_inference_batch: stacked, padded sequences; each mel equals the single call)
    import ctypes as C
    U = 8
    lib = ctx.lib
    lib.mia_flow_inference_batch.restype = C.c_int
    lib.mia_flow_inference_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_int]
    rep = lambda v: (C.c_void_p * U)(*([v] * U))
    bmels = [torch.empty_like(fmel) for _ in range(U)]
    outs = (C.c_void_p * U)(*[t.data_ptr() for t in bmels])
    n1, n2, n3 = np.full(U, n_tok, np.int32), np.full(U, n_prompt, np.int32), np.full(U, 2 * n_prompt, np.int32)

    def run_flow_batch():
        ctx.check(lib.mia_flow_inference_batch(fm.h, U, rep(tok.data_ptr()), n1.ctypes.data, rep(ptok.data_ptr()), n2.ctypes.data, rep(pfeat.data_ptr()),
                                               n3.ctypes.data, rep(spk.data_ptr()), rep(zz.data_ptr()), 0, outs, 1))

    run_flow_batch()
    e0.record()
    run_flow_batch()
    e1.record()
    torch.cuda.synchronize()
    msb = e0.elapsed_time(e1)
    res["flow_cosyvoice2_inference_batch8"] = {"mel_frames_per_s": round(U * n_new / (msb * 1e-3), 0), "ms": round(msb, 3), "utterances": U,
                                               "ms_per_utterance": round(msb / U, 3), "realtime_factor": round(U * n_new / 50.0 / (msb * 1e-3), 1),
                                               "equals_single_call": bool(all(torch.equal(t, fmel) for t in bmels))}
    fm.close()
    # ---- S3TokenizerV2 (CosyVoice2 prompt path): 10 s of 128-bin mel @ 100 Hz -> 250 speech tokens
    from mlx_swift_audio_amd import s3tok as HS
    scfg = S.S3_CONFIGS["s3_v2"]
    tk = HS.S3Tokenizer.load(ctx, scfg, S.s3_weights(scfg, 0))
    Ts = 1000
    smel = torch.randn(1, scfg.n_mels, Ts, device="cuda").contiguous()
    stoks = torch.zeros(1, Ts // 4 + 8, dtype=torch.int32, device="cuda")
    slen = np.asarray([Ts], np.int32)
    sn = np.zeros(1, np.int32)

    def run_s3():
        ctx.check(ctx.lib.mia_s3tok_encode(tk.h, smel.data_ptr(), slen.ctypes.data, 1, Ts, stoks.data_ptr(), stoks.shape[1], sn.ctypes.data, 1))

    run_s3()
    e0.record()
    for _ in range(reps):
        run_s3()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    res["s3tokenizer_v2_encode"] = {"tokens_per_s": round(int(sn[0]) / (ms * 1e-3), 0), "ms": round(ms, 3), "mel_frames": Ts, "tokens": int(sn[0]),
                                    "realtime_factor": round(Ts / 100.0 / (ms * 1e-3), 1), **alg(run_s3, ms)}
    tk.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="large-v3-turbo")
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--max-new-tokens", type=int, default=0, help="cap generated tokens per clip (0 = full 448 budget)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-codec", action="store_true", help="skip the SNAC/DAC decode samples/s side measurement")
    ap.add_argument("--cpu-tokens", type=int, default=200, help="greedy steps actually run by the CPU baseline (the rest of the 445-step budget is extrapolated linearly)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lm", default="orpheus-3b", help="LM decode leg (BASELINE configs[2] / [3]): Orpheus-3B on bf16 and packed MLX-q4 weights + CosyVoice2's Qwen2LM, each with its HBM roofline fraction")
    ap.add_argument("--no-lm", action="store_true", help="skip the LM leg (~1 min of checkpoint generation)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: only the N-rank launcher + the per-pass exchange order, on gloo (tests/test_multi_rank.py)")
    ap.add_argument("--lm-batch", type=int, default=32, help="sentences side by side in the batched part of --lm")
    ap.add_argument("--weight-sharing", choices=["auto", "off"], default="auto",
                    help="auto: concurrent replicas load the shared decode weights cacheable (mia_whisper_set_weight_sharing), the serial execution "
                         "streams them non-temporal; off: non-temporal everywhere (the round-2 behaviour)")
    ap.add_argument("--replicas", type=int, default=3,
                    help="model replicas on separate HIP streams; passes are dealt round-robin so the encoder of one batch overlaps the decoder of another (1 = strictly serial passes)")
    ap.add_argument("--phase-log", action="store_true", help="with --schedule phased: log the wall time of every encoder / decoder phase")
    ap.add_argument("--schedule", default="pipelined", choices=["pipelined", "phased"],
                    help="how R > 1 replicas share the GPU: pipelined = each replica runs whole passes on its own stream; phased = rounds of R passes, all encoders first, then all decoders concurrently")
    ap.add_argument("--no-config0", action="store_true", help="skip the BASELINE configs[0] leg (tiny.en, one 10 s clip: GPU + CPU at all cores and 1 thread)")
    ap.add_argument("--streams", default="prio", choices=["single", "prio"],
                    help="R > 1 replicas: single = each replica's whole pass on one stream; prio = decode chains on HIGH-priority streams, every replica's "
                         "encoder half on its own normal-priority stream (mia_whisper_set_encode_stream)")
    ap.add_argument("--dp", default="abi", choices=["abi", "torch"], help="token all-gather at N > 1: mia_dp_* (RCCL behind the C ABI, ONE communicator on the rank's exchange stream) or torch.distributed")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  This process starts N rank processes (one per GPU,
        # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and never touches a GPU itself -- counting devices does not
        # initialise the runtime -- so no exec happens after HIP initialisation anywhere.
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # rehearsal hook for a one-GPU box: every rank on device 0 and gloo for the (57 KB) token gather -- the N > 1 control flow without N GPUs
    rehearsal = os.environ.get("MIA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import mlx_swift_audio_amd as m
    from mlx_swift_audio_amd import synthetic as S
    from mlx_swift_audio_amd import whisper as HW

    dims = S.DIMS[args.model]
    dtype = m.BF16 if args.dtype == "bf16" else m.F16
    import threading
    from mlx_swift_audio_amd import parallel as P

    t0 = time.time()
    weights = S.synthetic_weights(dims, seed=args.seed, style="survey")
    log(f"[bench] synthetic {args.model} checkpoint generated in {time.time() - t0:.1f}s")
    B = args.batch
    clips = np.stack([S.synth_clip(rank * B + i) for i in range(B)])
    offs = np.arange(B + 1, dtype=np.int64) * clips.shape[1]
    main_stream = torch.cuda.Stream()     # non-default stream (graph capture is illegal on the legacy default stream)
    torch.cuda.set_stream(main_stream)
    pcm = torch.from_numpy(clips).cuda()
    torch.cuda.synchronize()

    class Replica:
        """One model instance on its own HIP stream.  Passes are dealt round-robin over R replicas from R host threads, so the
        (MFMA-bound) encoder of one batch overlaps the (launch-latency-bound) decoder of another -- how one GPU is driven for
        throughput.  Every pass is still a full log-mel + encode + 448-token-budget decode of 32 clips."""

        def __init__(self, root=None):
            prio = args.streams == "prio" and max(1, min(args.replicas, args.steps)) > 1
            self.stream = torch.cuda.Stream(priority=-1) if prio else torch.cuda.Stream()
            self.ctx = m.Context(local_rank, stream=self.stream.cuda_stream)
            # the first replica uploads the weights; the others are clones: own activations / KV caches / step graph, shared weights
            self.model = HW.WhisperModel.load(self.ctx, dims, weights, dtype) if root is None else root.model.clone(self.ctx)
            self.enc_stream = None
            if max(1, min(args.replicas, args.steps)) > 1 and args.weight_sharing == "auto":
                self.model.set_weight_sharing(max(1, min(args.replicas, args.steps)))
            if prio:       # the decode chain outranks the queued tiles of another batch's encoder (include/mia.h, mia_whisper_set_encode_stream)
                self.enc_stream = torch.cuda.Stream(priority=0)
                self.model.set_encode_stream(self.enc_stream.cuda_stream)
            self.opts = HW.DecodingOptions(suppress_ids=S.synthetic_suppress_list(self.model.special), blank_ids=[220], max_new_tokens=args.max_new_tokens)
            with torch.cuda.stream(self.stream):
                self.tokens = torch.zeros((B, self.opts.max_tokens), dtype=torch.int32, device="cuda")
                self.n_tok = torch.zeros(B, dtype=torch.int32, device="cuda")
                self.avg = torch.zeros(B, dtype=torch.float32, device="cuda")
                self.nsp = torch.zeros(B, dtype=torch.float32, device="cuda")
            self.stream.synchronize()

            self.pass_done = None      # event recorded behind the last decode on this replica's stream
            self.last_pass = -1        # id of the last pass this replica submitted to the exchange
            if world > 1:
                with torch.cuda.stream(self.stream):
                    self.all_tokens = torch.zeros((B * world, self.opts.max_tokens), dtype=torch.int32, device="cuda")
                    self.all_n = torch.zeros(B * world, dtype=torch.int32, device="cuda")
                self.stream.synchronize()

        def _before(self):
            # this replica's token buffers are about to be rewritten: its previous pass's gather must have consumed them (stream-side wait)
            if world > 1 and self.last_pass >= 0:
                exch.wait_issued(self.last_pass)
                self.stream.wait_event(self.gather_done)

        def step_encode(self):
            self.model.encode_windows_device(pcm.data_ptr(), offs)

        def step_decode(self, pass_id):
            self._before()
            self.model.decode_greedy_device(self.opts, self.tokens.data_ptr(), self.n_tok.data_ptr(), self.avg.data_ptr(), self.nsp.data_ptr())
            self._exchange(pass_id)

        def step(self, pass_id):
            self._before()
            self.model.transcribe_windows_device(pcm.data_ptr(), offs, self.opts, self.tokens.data_ptr(), self.n_tok.data_ptr(), self.avg.data_ptr(),
                                                 self.nsp.data_ptr())
            self._exchange(pass_id)

        def _exchange(self, pass_id):
            if world > 1:              # the path's only exchange, once per pass: handed to the rank's ONE exchange thread, which issues in pass order
                self.pass_done = torch.cuda.Event()
                self.pass_done.record(self.stream)
                self.gather_done = torch.cuda.Event()
                self.last_pass = pass_id
                exch.submit(pass_id, self)

    R = max(1, min(args.replicas, args.steps))
    t0 = time.time()
    reps = [Replica()]
    reps += [Replica(reps[0]) for _ in range(R - 1)]
    del weights
    log(f"[bench] {R} replica(s) loaded in {time.time() - t0:.1f}s")
    ctx = reps[0].ctx
    dp_mode = "none"
    exch = None
    pass_counter = [0]
    if world > 1:
        # ONE exchange channel per rank: its own stream + context + communicator; gathers issued by one thread in pass order (Exchanger)
        xstream = torch.cuda.Stream()
        xctx = m.Context(local_rank, stream=xstream.cuda_stream)
        use_abi = args.dp == "abi" and not rehearsal
        if use_abi:
            # mia_dp_init is a collective: vote on RCCL availability FIRST, so that a rank that cannot bind RCCL never leaves its peers
            # blocked inside ncclCommInitRank
            ok = torch.tensor([P.dp_available(xctx)], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            use_abi = int(ok.item()) == 1
            if not use_abi:
                log(f"[bench] rank {rank}: RCCL cannot be bound behind the C ABI on every rank; using torch.distributed for the gather")
        if use_abi:
            ids = [P.dp_unique_id(xctx) if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            P.dp_init(xctx, rank, world, ids[0])
            dp_mode = "mia_dp_gather_tokens (RCCL behind the C ABI): one communicator on the rank's exchange stream, gathers issued in pass order by one thread"
        else:
            dp_mode = "torch.distributed all_gather on the rank's exchange stream, issued in pass order by one thread"

        def issue(rp):
            xstream.wait_event(rp.pass_done)
            if use_abi:
                P.dp_gather_tokens(xctx, rp.tokens.data_ptr(), rp.n_tok.data_ptr(), B, rp.opts.max_tokens, B * world, rp.all_tokens.data_ptr(), rp.all_n.data_ptr())
            else:
                with torch.cuda.stream(xstream):
                    P.gather_tokens(rp.tokens, rp.n_tok, world, max_shard=B)
            rp.gather_done.record(xstream)

        exch = Exchanger(issue)

    def run_passes(k):
        """k passes dealt round-robin over the replicas (pass i -> replica i % R on EVERY rank), one host thread per replica; returns
        when every stream has drained."""
        base = pass_counter[0]
        pass_counter[0] += k
        if args.schedule == "phased" and R > 1:
            # rounds of up to R passes: every replica's log-mel + encoder first (MFMA-bound: they may share the chip freely), then every
            # replica's decoder concurrently from its own host thread.  Decoders of different batches overlap each other well (three
            # reach 1.5x the throughput of one, tools/decode_overlap_probe.py); an encoder next to a decoder does not (its 45 us tiles
            # hold every CU while the decoder's 46 short kernels per step queue behind them).
            left = k
            while left > 0:
                act = reps[:min(R, left)]
                tp0 = time.perf_counter()
                for rp in act:
                    rp.step_encode()
                for rp in act:
                    rp.ctx.synchronize()
                tp1 = time.perf_counter()
                ths = [threading.Thread(target=rp.step_decode, args=(base + (k - left) + i,)) for i, rp in enumerate(act)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                for rp in act:
                    rp.ctx.synchronize()
                    rp.stream.synchronize()
                if args.phase_log:
                    log(f"[bench] round of {len(act)}: encoders {1e3 * (tp1 - tp0):.1f} ms, decoders {1e3 * (time.perf_counter() - tp1):.1f} ms")
                left -= len(act)
        elif R == 1:
            for i in range(k):
                reps[0].step(base + i)
        else:
            ths = [threading.Thread(target=lambda rp=rp, r=r: [rp.step(base + i) for i in range(r, k, R)]) for r, rp in enumerate(reps) if r < k]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if exch is not None:
            exch.wait_issued(base + k - 1)
            xstream.synchronize()
        for rp in reps:
            rp.ctx.synchronize()
            rp.stream.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    prof_keys = ("logmel", "enc_gemm", "crosskv_gemm", "enc_attention", "enc_norm", "decode")
    run_passes(max(args.warmup, 1) * R if args.warmup else 0)     # every replica captures its step graph during warm-up
    fence()
    # ---- serial execution (replica 0 alone, instrumented): `roofline` and `stages` describe THIS execution -- HIP events on the
    # library's stream over whole passes with nothing else on the GPU, next to its own wall-clock throughput -- so that one set of
    # numbers describes one execution.  (Under R concurrent streams a launch's wall duration includes waiting for the other streams'
    # kernels and says nothing about the kernel; the pipelined timed region below therefore runs un-instrumented.)
    n_serial = 1 if R == 1 else 2
    serial_exec = None
    if R > 1:
        # a lone decode loop streams its weights non-temporal, concurrent loops on one weight copy load them cacheable
        # (mia_whisper_set_weight_sharing): the serial execution runs in the lone-loop form -- one untimed pass re-captures the step graph
        if args.weight_sharing == "auto":
            reps[0].model.set_weight_sharing(1)
        base = pass_counter[0]
        pass_counter[0] += 1
        reps[0].step(base)
        if exch is not None:
            exch.wait_issued(base)
            xstream.synchronize()
        fence()
        ctx.profile(True)
        ctx.profile_reset()
        fence()
        ts = time.perf_counter()
        base = pass_counter[0]
        pass_counter[0] += n_serial
        for i in range(n_serial):
            reps[0].step(base + i)
        if exch is not None:
            exch.wait_issued(base + n_serial - 1)
            xstream.synchronize()
        ctx.synchronize()
        reps[0].stream.synchronize()
        serial_s = time.perf_counter() - ts
        prof_serial = {k: ctx.profile_read(k) for k in prof_keys}
        serial_exec = {"replicas": 1, "passes": n_serial, "ms_per_step": round(serial_s / n_serial * 1e3, 3), "value": round(30.0 * B * n_serial / serial_s, 2), "unit": "audio-sec/s"}
        ctx.profile(False)
        if args.weight_sharing == "auto":
            reps[0].model.set_weight_sharing(R)
        run_passes(R)                                  # untimed: replica 0 re-captures its step graph in the shared-weights form
    else:
        ctx.profile(True)
        ctx.profile_reset()
    fence()
    t_start = time.perf_counter()
    run_passes(args.steps)
    fence()
    elapsed = time.perf_counter() - t_start
    ctx.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_gen = reps[0].n_tok.cpu().numpy()
    prof = {k: ctx.profile_read(k) for k in prof_keys} if R == 1 else prof_serial
    ms_per_step = elapsed / args.steps * 1e3
    audio_s = 30.0 * B * world
    value = audio_s * args.steps / elapsed
    model = reps[0].model

    L, D, V, T = dims.n_text_layer, dims.n_text_state, dims.n_vocab, dims.n_audio_ctx
    # weights touched once per step = per layer q|k|v|o (4 D^2) + cross q|o (2 D^2; cross k|v are applied once per clip by the
    # encode call, not per step) + MLP (8 D^2) = 14 D^2, + the tied embedding V D; + the cross-KV of every clip
    dec_bytes_per_step = 2.0 * (L * 14 * D * D + V * D) + B * L * 2 * T * D * 2.0

    def summarize(pf, passes):
        """roofline of the dominant kernel class (largest device time) + per-class figures, from HIP-event records of `passes` passes"""
        n_l, ms_l, w_l = pf["enc_gemm"]
        gemm_tflops = (w_l / n_l) / ((ms_l / n_l) * 1e-3) / 1e12 if n_l else 0.0
        n_d, ms_d, steps_d = pf["decode"]
        dec_gbs = dec_bytes_per_step * steps_d / (ms_d * 1e-3) / 1e9 if ms_d else 0.0
        tot = sum(v[1] for v in pf.values()) or 1.0
        if pf["enc_gemm"][1] >= pf["decode"][1]:
            rl = {"kernel": "gemm_nt_kernel_8ph (encoder Linear/Conv GEMMs, 256x256x64 MFMA tiles)", "bound": "mfma",
                  "achieved": round(gemm_tflops, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(gemm_tflops / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                  "launches": n_l, "avg_launch_ms": round(ms_l / max(n_l, 1), 4), "flop_per_launch_avg": w_l / max(n_l, 1)}
        else:
            rl = {"kernel": "decode step graph (skinny MFMA GEMMs + KV-cache attention + decode head)", "bound": "hbm",
                  "achieved": round(dec_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dec_gbs / HBM_PEAK_GBS, 4),
                  "traffic": None, "launches": int(steps_d), "avg_launch_ms": round(ms_d / max(steps_d, 1), 4),
                  "bytes_per_launch": dec_bytes_per_step}
        # HBM traffic per launch from the committed PMC passes of this same workload (rocprofv3 cannot run inside the timed process):
        # profiles/r01_pmc_summary.json, made by tools/pmc_summary.py with the guide's gfx950 corrections.  null when absent.
        # The passes were taken on the headline workload (large-v3-turbo, 32 clips): any other model / batch reports null.
        try:
            if args.model != "large-v3-turbo" or B != 32:
                raise KeyError("no PMC passes for this workload")
            pmc_name = next(n for n in ("r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            rl["traffic"] = round(pmc["decode_step"]["hbm_bytes_per_step"] if rl["bound"] == "hbm" else pmc["encoder_gemm"]["hbm_bytes_per_launch"], 0)
            rl["traffic_source"] = f"profiles/{pmc_name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2)"
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        stg = {k: {"ms_per_pass": round(v[1] / passes, 3), "share_of_device_time": round(v[1] / tot, 4)} for k, v in pf.items()}
        stg["enc_gemm"]["tflops"] = round(gemm_tflops, 1)
        if pf["enc_attention"][0]:
            stg["enc_attention"]["tflops"] = round(pf["enc_attention"][2] / (pf["enc_attention"][1] * 1e-3) / 1e12, 1)
        if pf["logmel"][0]:
            stg["logmel"]["GBs_algorithmic"] = round(pf["logmel"][2] / (pf["logmel"][1] * 1e-3) / 1e9, 1)
        stg["decode"]["GBs_algorithmic"] = round(dec_gbs, 1)
        stg["decode"]["steps_per_pass"] = steps_d / max(n_d, 1)
        return rl, stg

    roofline, stage = summarize(prof, args.steps if R == 1 else n_serial)
    roofline["measured_over"] = ("the timed region (serial passes)" if R == 1 else
                                 f"{n_serial} serial passes on one replica in this process, nothing else on the GPU (`execution`); the pipelined timed region of `value` runs {R} concurrent streams un-instrumented")
    if serial_exec:
        roofline["execution"] = serial_exec
    steps_d, n_d = prof["decode"][2], prof["decode"][0]
    if R > 1:      # device-level view of the timed region: algorithmic decode bytes of every pass over the wall time (encoders run in the same time)
        roofline["timed_region_decode_bytes_over_wall_GBs"] = round(dec_bytes_per_step * (steps_d / max(n_d, 1)) * args.steps / elapsed / 1e9, 1)

    out = {
        "metric": "audio-sec/s (Whisper large-v3-turbo b=32) at 1/2/4/8 GPU; codec samples/s",
        "value": round(value, 2), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"Whisper {args.model} {args.dtype}, batch={B}x30 s synthetic clips per MI355X, log-mel+encode+greedy decode "
                               f"(T=0, timestamps, max_tokens 448, one window per clip), random-init N(0,0.02^2) weights",
                   "clips_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "generated_tokens_per_clip_mean": float(n_gen.mean()), "decoder_steps_per_pass": steps_d / max(n_d, 1),
                   "realtime_factor": round(value, 1), "replicas": R, "exchange": dp_mode,
                   "schedule": args.schedule if R > 1 else "serial",
                   "pipeline": ("strictly serial passes" if R == 1 else
                                f"{R} model replicas (one weight copy, own activations / KV caches / step graphs) on {R} HIP streams, " +
                                ("rounds of up to R passes: all encoders, then all decoders concurrently (decoders of different batches overlap each other; an encoder beside a decoder does not)"
                                 if args.schedule == "phased" else
                                 "passes dealt round-robin: the encoder of one batch overlaps the decoder of another") +
                                "; the decode steps load the shared weights cacheable (mia_whisper_set_weight_sharing; the serial `execution` runs the lone-loop, non-temporal form)" +
                                "; every pass is a complete log-mel + encode + decode of its 32 clips")},
        "roofline": roofline, "stages": stage,
    }
    if rank == 0 and not args.no_codec:
        torch.cuda.set_stream(reps[0].stream)     # the codec leg times library work with torch events: same stream as the context
        out["codec"] = codec_bench(ctx, torch)
    if rank == 0 and args.lm and not args.no_lm:
        try:
            out["lm"] = lm_bench(ctx, torch, args.lm, args.lm_batch)
        except Exception as e:   # the side legs must never take the headline number down with them
            out["lm"] = {"error": repr(e)}
    if rank == 0 and not args.no_config0 and not args.no_cpu_baseline:
        try:
            out["config0"] = config0_leg(ctx, torch)
        except Exception as e:
            out["config0"] = {"error": repr(e)}
    if rank == 0 and not args.no_cpu_baseline:
        try:
            ncpu = min(16, len(os.sched_getaffinity(0)))   # the GPU box's CPU share for one GPU is 16 cores
            out["cpu_baseline"] = cpu_baseline(args.model, args.seed, ncpu, args.cpu_tokens)
        except Exception as e:  # the CPU leg must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "unit": "audio-sec/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
    if world > 1:
        exch.close(pass_counter[0])
        if use_abi:
            P.dp_shutdown(xctx)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    for rp in reversed(reps):                     # clones before the replica that owns the weights
        rp.model.close()


if __name__ == "__main__":
    main()
