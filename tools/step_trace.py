#!/usr/bin/env python3
"""Per-kernel durations of ONE decode step from a rocprofv3 --kernel-trace CSV (tools/step_trace.py <kernel_trace.csv> [step_from_end]).
Inside a hipGraph the timestamps are contiguous (a node's duration includes its boundary), so the sum is the step."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void dec_head_final")]
i0, i1 = idx[-k - 1], idx[-k]
tot = 0
for j in range(i0 + 1, i1 + 1):
    r = rows[j]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    g = int(r["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])
    tot += d
    print(f"{r['Kernel_Name'][:64]:64s} {d / 1e3:7.2f} us  gap {g / 1e3:5.2f}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
print(f"sum of durations {tot / 1e3:.1f} us; step (end to end) {(int(rows[i1]['End_Timestamp']) - int(rows[i0]['End_Timestamp'])) / 1e3:.1f} us")
