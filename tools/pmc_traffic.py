#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_<w>_FETCH_SIZE -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_<w>_WRITE_SIZE -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_<w>_FETCH_SIZE gpurun_out/pmc_<w>_WRITE_SIZE [commit [clips_per_step]] > profiles/rN_pmc_traffic_<w>.json
(the profiled command is `bench.py --profile-only-batch`: warm-up + timed steps of one workload, nothing else; the
optional third argument stamps the JSON with the commit the library was built from)

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads (16 B/lane) -> doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (narrower stores are uncalibrated, kept as is)."""
import glob
import json
import re
import sys

import pandas as pd

# (the fused pair-conv kernels belong to the gemm class, like their ProfScope in api_forward.hip; kernel names carry their
#  parameter types -- pair_conv_fused_kernel(..., EpiConvT16, EpiUnionT16) -- hence the explicit exclusion under union_conv)
CLASSES = [("gemm", r"gemm(_sk|16c?)_kernel<.*(EpiLinear|EpiHeads|EpiConvRelBn|EpiConvT16)|gemm(16c?)?_fixup(_vec)?_kernel<.*(EpiLinear|EpiHeads|EpiConvRelBn|EpiConvT16)|pair_conv_fused"),
           ("union_conv", r"^(?!.*pair_conv_fused).*EpiUnion"), ("attention", r"attention"), ("layernorm", r"layernorm"),
           ("mask_conv", r"mask_conv1_pool"), ("index", r"pair_prep|gather_rows|objcls")]


def load(d, counter):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    t = pd.read_csv(f)
    t = t[t["Counter_Name"] == counter]
    return t[["Dispatch_Id", "Kernel_Name", "Counter_Value"]]


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name, pat in CLASSES:
        f = fetch[fetch["Kernel_Name"].str.contains(pat, regex=True)]
        w = write[write["Kernel_Name"].str.contains(pat, regex=True)]
        if not len(f):
            continue
        main_launch = f[~f["Kernel_Name"].str.contains("fixup")]
        n = len(main_launch)
        rd = float(f["Counter_Value"].sum()) * 1024 * 2       # gfx950: FETCH_SIZE = 1/2 of streamed bytes
        wr = float(w["Counter_Value"].sum()) * 1024
        out[name] = {"launches": int(n), "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / n,
                     "hbm_bytes_per_launch": (rd + wr) / n}
    json.dump({"note": "FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per launch (fix-up launches folded into "
                       "their GEMM); all forwards of the profiled bench run (bench.py --profile-only-batch)",
               "commit": sys.argv[3] if len(sys.argv) > 3 else None,
               # clips per step of the profiled run (bench.py attaches `traffic` only to a run of the same batch)
               "clips_per_step": int(sys.argv[4]) if len(sys.argv) > 4 else None,
               "classes": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
