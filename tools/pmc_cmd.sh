#!/bin/bash
# PMC counters of an arbitrary python3 command, one counter per rocprofv3 pass (the pool refuses combinations):
#     bash tools/pmc_cmd.sh <out prefix> "<kernel pattern> [<pattern> ...]" <python script> [args...]
# -> gpurun_out/<prefix>_pmc.json (tools/pmc_kernels.py: per-kernel means + derived ratios).  python3 sits directly behind `--`.
set -euo pipefail
P=${1:?prefix}; PATS=${2:?kernel name patterns}; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS; do
  rm -rf "$O/${P}_pmcx_$c"
  rocprofv3 --pmc "$c" --kernel-trace --output-format csv -d "$O/${P}_pmcx_$c" -- python3 "$@" > /dev/null 2> "$O/${P}_pmcx_$c.err" || echo "pass $c failed"
done
# shellcheck disable=SC2086
python3 tools/pmc_kernels.py "$O/${P}_pmcx_" manual $PATS > "$O/${P}_pmc.json"
rm -rf "$O/${P}"_pmcx_*/
find "$O" -maxdepth 1 -name "${P}_pmcx_*.err" -size -1k -delete
cat "$O/${P}_pmc.json"
