#!/bin/bash
# Does a start-up skew between the workgroups that share operand panels cut the fabric reads of the 128 x 176 GEMM?
# (experiment build of the library as $1; STTRAN_T16_SKEW units of ~1 024 clocks; FETCH_SIZE x 2 per the gfx950 note)
set -uo pipefail
LIB=${1:?experiment build of libsttran_hip.so}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export STTRAN_LIB=$LIB
for shape in 21120,1936,1936 21120,5808,1936; do
  for skew in 0 1 2 4; do
    export STTRAN_T16_SKEW=$skew
    tf=$(python3 tools/gemm_bench.py --one $shape --tiles 5 --residual --iters 20 2>/dev/null | tail -1)
    rm -rf gpurun_out/skew_pmc
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/skew_pmc -- python3 tools/gemm_bench.py --one $shape --tiles 5 --residual --iters 3 > /dev/null 2>&1
    rd=$(python3 - <<'PY'
import glob
import pandas as pd
t = pd.read_csv(glob.glob("gpurun_out/skew_pmc/*/*counter_collection.csv")[0])
t = t[(t["Counter_Name"] == "FETCH_SIZE") & t["Kernel_Name"].str.contains("gemm16_kernel")]
print(f"{t['Counter_Value'].tail(3).mean() * 2048 / 1e6:.0f} MB read per launch")
PY
)
    echo "shape $shape skew $skew: $tf | $rd"
  done
done
rm -rf gpurun_out/skew_pmc
