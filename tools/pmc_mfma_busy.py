#!/usr/bin/env python3
"""MFMA-pipe occupancy of one kernel from rocprofv3 PMC passes (one counter per pass, as the pool requires):

    for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; do
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcm_$c -- python3 bench.py --profile-only-batch --steps 3 --warmup 1
    done
    python tools/pmc_mfma_busy.py gpurun_out/pmcm_ "GemmTile<256, 128, 4, 2, 3>, sttran::EpiLinearV" [commit] > profiles/rN_pmc_mfma_busy.json

mfma_busy_fraction = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): the share of the kernel's
time its matrix pipes were busy, padding MFMAs included (rocprofv3 sums SQ counters over the SIMDs and GRBM over the
XCDs, MI355X_MICROARCH.md 'DVFS give-back')."""
import glob
import json
import sys

import pandas as pd

COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]


def main():
    prefix, pattern = sys.argv[1], sys.argv[2]
    out = {}
    launches = None
    for c in COUNTERS:
        fs = glob.glob(f"{prefix}{c}/*/*counter_collection.csv")
        if not fs:
            continue
        t = pd.read_csv(fs[0])
        t = t[(t["Counter_Name"] == c) & t["Kernel_Name"].str.contains(pattern, regex=False) & ~t["Kernel_Name"].str.contains("fixup")]
        if len(t):
            out[c] = float(t["Counter_Value"].mean())
            launches = len(t)
    d = {}
    # shader clock of the PROFILED launches: GRBM_GUI_ACTIVE per XCD / kernel duration (profiled passes run at a lower
    # clock than unprofiled ones: MI355X_MICROARCH.md)
    fs = glob.glob(f"{prefix}GRBM_GUI_ACTIVE/*/*kernel_trace.csv")
    if fs and "GRBM_GUI_ACTIVE" in out:
        k = pd.read_csv(fs[0])
        k = k[k["Kernel_Name"].str.contains(pattern, regex=False) & ~k["Kernel_Name"].str.contains("fixup")]
        if len(k):
            dur_ns = float((k["End_Timestamp"] - k["Start_Timestamp"]).mean())
            d["profiled_kernel_us"] = dur_ns / 1e3
            d["shader_clock_ghz_profiled"] = (out["GRBM_GUI_ACTIVE"] / 8.0) / dur_ns
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
        d["mfma_busy_fraction"] = (out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (out["GRBM_GUI_ACTIVE"] / 8.0)
    if "SQ_WAVE_CYCLES" in out:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in out:
                d[k.lower() + "_fraction_of_wave_cycles"] = out[k] / out["SQ_WAVE_CYCLES"]
    json.dump({"note": "rocprofv3 --pmc <counter> --kernel-trace, one counter per pass, python3 bench.py --profile-only-batch "
                       f"--steps 3 --warmup 1; mean per launch over {launches} launches of the kernel matching {pattern!r}",
               "commit": sys.argv[3] if len(sys.argv) > 3 else None, **out, "derived": d}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
