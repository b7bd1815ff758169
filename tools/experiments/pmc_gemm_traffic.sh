#!/bin/bash
# Fabric traffic (FETCH_SIZE x 2 per the gfx950 note, WRITE_SIZE) of ONE GEMM shape per tile: pmc_gemm_traffic.sh M,N,K tiles tag
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
shape=$1; tiles=$2; out=$3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "gpurun_out/${out}_$c"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "gpurun_out/${out}_$c" -- python3 tools/gemm_bench.py --one "$shape" --tiles "$tiles" --residual --iters 3 > /dev/null 2> "gpurun_out/${out}_$c.err" || true
done
python3 - "$out" <<'PY'
import glob, sys
import pandas as pd
out = sys.argv[1]
res = {}
for c in ["FETCH_SIZE", "WRITE_SIZE"]:
    t = pd.read_csv(glob.glob(f"gpurun_out/{out}_{c}/*/*counter_collection.csv")[0])
    t = t[(t["Counter_Name"] == c) & t["Kernel_Name"].str.contains("gemm") & ~t["Kernel_Name"].str.contains("fixup")]
    for k, g in t.groupby("Kernel_Name"):
        v = g["Counter_Value"].tail(3).mean() * 1024 * (2 if c == "FETCH_SIZE" else 1)
        res.setdefault(k[:90], {})[c] = v
for k, v in res.items():
    print(f"{k}: read {v.get('FETCH_SIZE', 0) / 1e6:.0f} MB  write {v.get('WRITE_SIZE', 0) / 1e6:.0f} MB per launch")
PY
