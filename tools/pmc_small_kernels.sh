#!/bin/bash
# PMC passes (one counter each) of one --profile-only-batch run, reduced for the NON-GEMM kernels of the step:
#     gpurun -- bash tools/pmc_small_kernels.sh r5_a $(git rev-parse --short HEAD)
set -euo pipefail
P=${1:?prefix}; C=${2:?commit}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS; do
  rm -rf "$O/${P}_pmcs_$c"
  rocprofv3 --pmc "$c" --kernel-trace --output-format csv -d "$O/${P}_pmcs_$c" -- python3 bench.py --steps 3 --warmup 1 --profile-only-batch \
    > /dev/null 2> "$O/${P}_pmcs_$c.err" || echo "pass $c failed"
done
python3 tools/pmc_kernels.py "$O/${P}_pmcs_" "$C" mask_conv1_pool_kernel attention_short_kernel layernorm_kernel EpiUnionT16 EpiConvT16 "Tile16<128, 176>" > "$O/${P}_pmc_kernels.json"
rm -rf "$O/${P}"_pmcs_*/
cat "$O/${P}_pmc_kernels.json"
