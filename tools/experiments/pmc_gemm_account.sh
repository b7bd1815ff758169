#!/bin/bash
# Fabric-traffic account of the dominant GEMM kernels: per shape FETCH_SIZE (x 2: gfx950 note), WRITE_SIZE,
# TCC_HIT_sum, TCC_MISS_sum (one counter per rocprofv3 pass) of tools/gemm_bench.py --one M,N,K on the tile the planner uses,
# next to the algorithmic bytes.   gpurun -- bash tools/experiments/pmc_gemm_account.sh r5 <commit>
set -uo pipefail
P=${1:?prefix}; C=${2:?commit}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
SHAPES=${SHAPES:-"21120,1936,1936,5,1 10560,1936,1936,5,1 21120,5808,1936,5,0 10560,5808,1936,5,0 21120,2048,1936,7,0 11264,512,12544,7,0"}
export SHAPES
for sh in $SHAPES; do
  IFS=, read M N K T R <<< "$sh"
  res=""; [ "$R" = 1 ] && res="--residual"
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
    d="$O/${P}_acct_${M}_${N}_${K}_$c"; rm -rf "$d"
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -- python3 tools/gemm_bench.py --one "$M,$N,$K" --tiles "$T" $res --iters 3 \
      > "$d.out" 2> "$d.err" || echo "pass $sh $c failed"
  done
done
python3 - "$P" "$C" <<'PY'
import glob, json, sys
import pandas as pd
P, C = sys.argv[1], sys.argv[2]
rows = []
import os
for sh in os.environ["SHAPES"].split():
    M, N, K, T, R = (int(v) for v in sh.split(","))
    rec = {"M": M, "N": N, "K": K, "tile": {0: "planner's choice", 5: "128x176", 7: "128x128 (16x16x4)"}.get(T, str(T)), "residual": bool(R)}
    for c in ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]:
        fs = glob.glob(f"gpurun_out/{P}_acct_{M}_{N}_{K}_{c}/*/*counter_collection.csv")
        if not fs:
            continue
        t = pd.read_csv(fs[0])
        t = t[(t["Counter_Name"] == c) & t["Kernel_Name"].str.contains("gemm16_kernel") & ~t["Kernel_Name"].str.contains("fixup")]
        if len(t):
            rec[c] = float(t["Counter_Value"].tail(3).mean())
    tfl = [l for l in open(f"gpurun_out/{P}_acct_{M}_{N}_{K}_FETCH_SIZE.out").read().splitlines() if " TF" in l and "auto[" in l]
    rec["bench_line"] = tfl[-1].strip() if tfl else None
    a, w, c_ = M * K * 4, N * K * 4, M * N * 4
    rec["algorithmic_read_bytes"] = a + w + (c_ if R else 0)
    rec["algorithmic_write_bytes"] = c_
    if "FETCH_SIZE" in rec:
        rec["fabric_read_bytes"] = rec["FETCH_SIZE"] * 1024 * 2          # KiB, x 2 per MI355X_MICROARCH.md (128-byte requests tallied at 64)
        rec["read_amplification"] = rec["fabric_read_bytes"] / rec["algorithmic_read_bytes"]
    if "WRITE_SIZE" in rec:
        rec["fabric_write_bytes"] = rec["WRITE_SIZE"] * 1024
        rec["write_amplification"] = rec["fabric_write_bytes"] / rec["algorithmic_write_bytes"]
    if "TCC_HIT_sum" in rec and "TCC_MISS_sum" in rec:
        rec["l2_hit_rate"] = rec["TCC_HIT_sum"] / (rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"])
    rows.append(rec)
json.dump({"note": "tools/experiments/pmc_gemm_account.sh: gemm_bench --one M,N,K (3 timed launches, back to back: operands of "
                   "< 256 MB may be served by the Infinity Cache, which the fabric counters still count); mean of the last 3 launches",
           "commit": C, "shapes": rows}, open(f"gpurun_out/{P}_gemm_traffic_account.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
rm -rf "$O/${P}"_acct_*/
