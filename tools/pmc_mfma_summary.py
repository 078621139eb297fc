"""MFMA utilisation per kernel from one rocprofv3 PMC pass -> profiles/<round>_pmc_mfma_summary.json.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d <dir> -o m \
        -- python3 bench.py --steps 1 --warmup 1 --max-new-tokens 4 --no-cpu-baseline --no-codec --replicas 1
    python tools/pmc_mfma_summary.py <dir> <out.json>

ROCm 7.2 ships no derived `MfmaUtil` for gfx950 (MI355X_MICROARCH.md, "rocprofv3 PMC slots"), so it is formed here:
SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of the cycles an MFMA occupies the matrix pipe (16 per v_mfma_f32_16x16x32_bf16: checked
against the FLOP count below), GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in cc:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    dur = defaultdict(float)
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    kernels = {}
    for k, v in agg.items():
        mf, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if mf <= 0 or gui <= 0:
            continue
        kernels[k] = {"launches": len(launches[k]), "mfma_busy_cycles": mf, "gui_active_cycles_sum_over_xcds": gui,
                      "total_ms": dur.get(k, 0.0) / 1e6, "clock_GHz_while_active": gui / 8.0 / dur[k] if dur.get(k) else None,
                      "mfma_util": mf / (gui / 8.0 * 1024.0)}
    enc = {k: v for k, v in kernels.items() if "gemm_nt_kernel" in k}
    tot_mf = sum(v["mfma_busy_cycles"] for v in enc.values())
    tot_gui = sum(v["gui_active_cycles_sum_over_xcds"] for v in enc.values())
    res = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -- "
                      "python3 bench.py --steps 1 --warmup 1 --max-new-tokens 4 --no-cpu-baseline --no-codec --replicas 1",
           "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
           "encoder_gemm_mfma_util": tot_mf / (tot_gui / 8.0 * 1024.0) if tot_gui else None,
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"]))}
    json.dump(res, open(out, "w"), indent=1)
    print("encoder GEMM MfmaUtil:", res["encoder_gemm_mfma_util"])
    for k, v in list(res["kernels"].items())[:6]:
        print(f'{v["mfma_util"]:.3f}  {v["clock_GHz_while_active"]:.2f} GHz  {k[:90]}')


if __name__ == "__main__":
    main()
