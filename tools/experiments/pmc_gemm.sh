#!/bin/bash
# MFMA-pipe occupancy of ONE GEMM shape on chosen tiles: rocprofv3 --pmc, one counter per pass (the pool refuses more),
# python3 directly behind `--`.  usage: pmc_gemm.sh M,N,K tiles out_prefix      (run on the GPU box, from the repo root)
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
shape=$1; tiles=$2; out=$3
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES; do
  rm -rf "gpurun_out/${out}_$c"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "gpurun_out/${out}_$c" -- python3 tools/gemm_bench.py --one "$shape" --tiles "$tiles" --residual --iters 3 > /dev/null 2> "gpurun_out/${out}_$c.err" || true
done
python3 - "$out" <<'PY'
import glob, sys
import pandas as pd
out = sys.argv[1]
vals = {}
for c in ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"]:
    fs = glob.glob(f"gpurun_out/{out}_{c}/*/*counter_collection.csv")
    if not fs:
        continue
    t = pd.read_csv(fs[0])
    t = t[(t["Counter_Name"] == c) & ~t["Kernel_Name"].str.contains("fixup") & t["Kernel_Name"].str.contains("gemm")]
    for k, g in t.groupby("Kernel_Name"):
        vals.setdefault(k[:110], {})[c] = (float(g["Counter_Value"].mean()), len(g))
for k, v in vals.items():
    line = k
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        line += f"  mfma_busy={(v['SQ_VALU_MFMA_BUSY_CYCLES'][0] / 1024) / (v['GRBM_GUI_ACTIVE'][0] / 8):.3f}"
    if "SQ_WAIT_INST_ANY" in v and "SQ_WAVE_CYCLES" in v:
        line += f"  wait_inst_any={v['SQ_WAIT_INST_ANY'][0] / v['SQ_WAVE_CYCLES'][0]:.3f}"
    if "GRBM_GUI_ACTIVE" in v:
        line += f"  gui_cycles_per_xcd={v['GRBM_GUI_ACTIVE'][0] / 8:.0f} launches={v['GRBM_GUI_ACTIVE'][1]}"
    print(line)
PY
