"""Aggregate rocprofv3 --pmc passes into profiles/<round>_pmc_summary.json.

    python tools/pmc_summary.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json>

Each pass was collected in its own run (TCC counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 1 --warmup 1 --max-new-tokens 32 ...
Corrections (MI355X_MICROARCH.md, HBM section): Counter_Value is in KiB-like units of 1024 B as printed by rocprofv3; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads -> doubled here; WRITE_SIZE is taken as is.  Values are per launch (means)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    rows = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            rows[k][0] += 1
            rows[k][1] += float(r["Counter_Value"]) * 1024.0
    return rows


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    # one code revision per summary: both passes must have launched exactly the same kernels the same number of times (a directory that
    # accumulated runs of different builds, or two passes of different commits, would average unrelated kernels into one figure)
    fk, wk = {k: v[0] for k, v in fetch.items()}, {k: v[0] for k, v in write.items()}
    if fk != wk:
        diff = sorted(set(fk.items()) ^ set(wk.items()))[:6]
        raise SystemExit(f"pmc_summary: the FETCH and WRITE passes do not cover the same launches (first differences: {diff}); re-take both on one build into fresh directories")
    heads = [k for k in fk if k.startswith("dec_head")]
    if len({fk[k] for k in heads}) > 1 or any(k.startswith("dec_head(") for k in heads) and any(k.startswith("dec_head_partial") for k in heads):
        raise SystemExit(f"pmc_summary: mixed decode-head kernels {[(k[:40], fk[k]) for k in heads]}: more than one build in these directories")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        n = fetch.get(k, [0, 0])[0] or write.get(k, [0, 0])[0]
        fb = 2.0 * fetch.get(k, [0, 0.0])[1]
        wb = write.get(k, [0, 0.0])[1]
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1)}
    # decode step = every dec_* kernel between two dec_head launches
    steps = max([v["launches"] for k, v in kernels.items() if k.startswith("dec_head")] or [0])      # one dec_head_partial (or dec_head) launch per step
    dec_f = sum(v["fetch_bytes_per_launch"] * v["launches"] for k, v in kernels.items() if "dec_" in k)
    dec_w = sum(v["write_bytes_per_launch"] * v["launches"] for k, v in kernels.items() if "dec_" in k)
    gem = {k: v for k, v in kernels.items() if "gemm_nt_kernel" in k}
    gl = sum(v["launches"] for v in gem.values())
    import subprocess
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except OSError:
        head = ""
    res = {
        "git_head_when_summarised": head,
        "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --max-new-tokens 32 --no-cpu-baseline --no-codec",
        "corrections": "FETCH_SIZE x2 (gfx950 wide-read undercount), WRITE_SIZE as reported; bytes = Counter_Value x 1024",
        "decode_step": {"steps": steps, "fetch_bytes_per_step": dec_f / max(steps, 1), "write_bytes_per_step": dec_w / max(steps, 1),
                        "hbm_bytes_per_step": (dec_f + dec_w) / max(steps, 1)},
        "encoder_gemm": {"launches": gl,
                         "hbm_bytes_per_launch": sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in gem.values()) / max(gl, 1)},
        "kernels": kernels,
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["decode_step"]), json.dumps(res["encoder_gemm"]))


if __name__ == "__main__":
    main()
