#!/usr/bin/env python3
"""Instruction mix of every MFMA loop of a kernel, from hipcc's assembly -- the check that found the if-converted tail masks, the IEEE
division inside GELU and the libm sinf in the snake prologue.

    python tools/isa_loop_stats.py mlx-swift-audio_amd/csrc/attention.hip enc_attention_kernelI4BF16 [min_mfma]

Compiles the source for gfx950 with the build's flags (device side only), finds the kernel whose mangled name contains the given
substring, and prints for each backward branch whose body holds at least `min_mfma` MFMAs: instruction count, MFMA count and the most
frequent opcodes.  Counts are static: code under a branch the launch never takes is counted too (read the branches before concluding)."""
import re
import subprocess
import sys
import tempfile
from collections import Counter

src, pat = sys.argv[1], sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
with tempfile.NamedTemporaryFile(suffix=".s") as f:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", f.name, src],
                   check=True, stderr=subprocess.DEVNULL)
    s = open(f.name).read().split("\n")
starts = [i for i, l in enumerate(s) if re.match(r"^_Z\S+:", l) and pat in l]
if not starts:
    raise SystemExit(f"no kernel matching {pat!r}")
for start in starts:
    end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
    body = s[start:end]
    ops = lambda seg: Counter(x.strip().split(" ")[0] for x in seg if x.strip() and not x.strip().startswith((";", ".")))
    tot = ops(body)
    print(f"{s[start].split(':')[0]}\n  whole kernel: {sum(tot.values())} instructions, {sum(v for k, v in tot.items() if 'mfma' in k)} MFMA, "
          f"{tot['v_div_scale_f32']} v_div_scale, {sum(v for k, v in tot.items() if 'scratch' in k)} scratch ops")
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l.strip()))}
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            c = ops(body[labels[m.group(1)]:i])
            n_mfma = sum(v for k, v in c.items() if "mfma" in k)
            if n_mfma >= min_mfma:
                print(f"  loop -> {m.group(1)}: {sum(c.values())} instructions, {n_mfma} MFMA, {sum(v for k, v in c.items() if 'branch' in k)} branches")
                print("    " + ", ".join(f"{k} x{v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:14]))
