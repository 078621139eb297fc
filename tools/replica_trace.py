#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 --kernel-trace CSV of `bench.py --replicas R` (tools/replica_cliff.sh): per hardware queue the
busy time and the gaps between consecutive kernels, and over the decode phase how many queues have a kernel in flight at once.
usage: python tools/replica_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
dec = [r for r in rows if r["Kernel_Name"].startswith(("void dec_", "dec_"))]
t0, t1 = dec[len(dec) // 4]["s"], dec[3 * len(dec) // 4]["e"]          # the middle half of the decode kernels: steady state
win = [r for r in rows if r["e"] > t0 and r["s"] < t1]
byq = defaultdict(list)
for r in win:
    byq[r[qkey]].append(r)
print(f"window {1e-6 * (t1 - t0):.1f} ms, {len(win)} kernels, queues: {len(byq)} ({qkey})")
for q, ks in sorted(byq.items()):
    busy = sum(min(k["e"], t1) - max(k["s"], t0) for k in ks)
    gaps = sorted(b["s"] - a["e"] for a, b in zip(ks, ks[1:]))
    nd = sum(1 for k in ks if "dec_" in k["Kernel_Name"])
    if not gaps:
        continue
    print(f"  queue {q}: {len(ks)} kernels ({nd} decode), busy {100.0 * busy / (t1 - t0):.1f} %, gap median {gaps[len(gaps) // 2] / 1e3:.2f} us, "
          f"p90 {gaps[int(0.9 * len(gaps))] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us, mean kernel {busy / len(ks) / 1e3:.2f} us")
# concurrency histogram: sweep events
ev = []
for r in win:
    ev.append((max(r["s"], t0), 1, r[qkey]))
    ev.append((min(r["e"], t1), -1, r[qkey]))
ev.sort()
active = defaultdict(int)
hist = defaultdict(int)
last = t0
for t, d, q in ev:
    n = sum(1 for v in active.values() if v > 0)
    hist[n] += t - last
    last = t
    active[q] += d
tot = sum(hist.values()) or 1
print("  queues with a kernel in flight: " + ", ".join(f"{n}: {100.0 * v / tot:.1f} %" for n, v in sorted(hist.items())))
# per-kernel-class mean duration of the decode chain (how much each kernel stretches under concurrency)
cls = defaultdict(lambda: [0, 0])
for r in win:
    n = r["Kernel_Name"].split("(")[0][:48]
    cls[n][0] += 1
    cls[n][1] += r["e"] - r["s"]
for n, (c, d) in sorted(cls.items(), key=lambda x: -x[1][1])[:8]:
    print(f"    {n:48s} n {c:6d} mean {d / c / 1e3:7.2f} us")
