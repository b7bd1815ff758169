#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 PMC passes (ONE counter per pass, as the pool requires) for SEVERAL kernels at once:

    python tools/pmc_kernels.py <dir prefix> <commit> <pattern> [<pattern> ...] > profiles/rN_pmc_kernels.json

`<dir prefix><COUNTER>/` = output directory of `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d ... --
python3 bench.py --profile-only-batch --steps 3 --warmup 1`.  For every kernel-name pattern: mean counter value per
launch, the launch count, the mean profiled duration, and the derived ratios (MFMA busy share of the kernel's time, wait
shares of the wave cycles, LDS bank-conflict share of the LDS cycles)."""
import glob
import json
import sys

import pandas as pd


def main():
    prefix, commit, patterns = sys.argv[1], sys.argv[2], sys.argv[3:]
    dirs = sorted(glob.glob(prefix + "*/"))
    out = {"note": "rocprofv3 --pmc <counter> --kernel-trace, one counter per pass; mean per launch", "commit": commit, "kernels": {}}
    for pat in patterns:
        rec, launches = {}, None
        for d in dirs:
            c = d[len(prefix):].strip("/")
            fs = glob.glob(f"{d}*/*counter_collection.csv")
            if not fs:
                continue
            t = pd.read_csv(fs[0])
            t = t[(t["Counter_Name"] == c) & t["Kernel_Name"].str.contains(pat, regex=False) & ~t["Kernel_Name"].str.contains("fixup")]
            if len(t):
                rec[c] = float(t["Counter_Value"].mean())
                launches = len(t)
            kf = glob.glob(f"{d}*/*kernel_trace.csv")
            if kf and "profiled_kernel_us" not in rec:
                k = pd.read_csv(kf[0])
                k = k[k["Kernel_Name"].str.contains(pat, regex=False) & ~k["Kernel_Name"].str.contains("fixup")]
                if len(k):
                    rec["profiled_kernel_us"] = float((k["End_Timestamp"] - k["Start_Timestamp"]).mean()) / 1e3
        dv = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and "GRBM_GUI_ACTIVE" in rec:
            dv["mfma_busy_fraction"] = (rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (rec["GRBM_GUI_ACTIVE"] / 8.0)
        if "GRBM_GUI_ACTIVE" in rec and "profiled_kernel_us" in rec:
            dv["shader_clock_ghz_profiled"] = rec["GRBM_GUI_ACTIVE"] / 8.0 / (rec["profiled_kernel_us"] * 1e3)
        if "SQ_WAVE_CYCLES" in rec:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
                if k in rec:
                    dv[k.lower() + "_fraction_of_wave_cycles"] = rec[k] / rec["SQ_WAVE_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in rec and "SQ_LDS_IDX_ACTIVE" in rec and rec["SQ_LDS_IDX_ACTIVE"]:
            dv["lds_bank_conflict_fraction_of_lds_cycles"] = rec["SQ_LDS_BANK_CONFLICT"] / rec["SQ_LDS_IDX_ACTIVE"]
        out["kernels"][pat] = {"launches": launches, **rec, "derived": dv}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
