#!/bin/bash
# Profiles of one round, taken on the GPU box at the commit the snapshot was made from:
#     gpurun -- bash tools/profile_round.sh r3_f $(git rev-parse --short HEAD)
# (the box has no .git: the commit is an argument, and it is stamped into every JSON this script derives).
# Every rocprofv3 pass is `--kernel-trace` with at most ONE `--pmc` counter (the pool refuses other combinations) and
# has python3 directly behind `--`.  Outputs go to gpurun_out/<prefix>_*; copy the summaries to profiles/.
set -euo pipefail
P=${1:?prefix, e.g. r3_f}
C=${2:?commit the library was built from}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
kt() {   # kt <name> <bench args...>: kernel trace + stats of one --profile-only-batch run -> <name>_kernel_stats.csv
  local name=$1; shift
  rm -rf "$O/${P}_kt_$name"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${P}_kt_$name" -- python3 bench.py "$@" --profile-only-batch \
    > "$O/${P}_kt_$name.json" 2> "$O/${P}_kt_$name.err"
  cp "$O/${P}_kt_$name"/*/*kernel_stats.csv "$O/${P}_${name}_kernel_stats.csv"
}
pmc() {  # pmc <dir name> <counter> <bench args...>
  local name=$1 ctr=$2; shift 2
  rm -rf "$O/${P}_pmc_${name}_$ctr"
  rocprofv3 --pmc "$ctr" --kernel-trace --output-format csv -d "$O/${P}_pmc_${name}_$ctr" -- python3 bench.py "$@" --profile-only-batch \
    > /dev/null 2> "$O/${P}_pmc_${name}_$ctr.err"
}
# 1. kernel traces: per-step averages of the three workloads of the line, the single clip, the second model
kt 16x12 --workload 16x12 --steps 20 --warmup 3
kt 64x36 --workload 64x36 --steps 10 --warmup 3
kt single --clips-per-step 1 --steps 50 --warmup 3
kt single_64x36 --workload 64x36 --clips-per-step 1 --steps 20 --warmup 3
kt dsgdetr_16x12 --model dsgdetr --steps 10 --warmup 3
kt dsgdetr_64x36 --model dsgdetr --workload 64x36 --steps 10 --warmup 3
kt 16x12_bf16x3 --workload 16x12 --gemm-engine bf16x3 --steps 20 --warmup 3       # the second engine (opt-in; never `value`)
# 2. fabric traffic of the GEMM class (FETCH_SIZE x 2 per the gfx950 note, WRITE_SIZE), STTran and DSG-DETR
for w in 16x12 64x36; do
  for c in FETCH_SIZE WRITE_SIZE; do pmc "$w" $c --workload $w --steps 3 --warmup 1; done
  cps=64; [ $w = 64x36 ] && cps=4
  python3 tools/pmc_traffic.py "$O/${P}_pmc_${w}_FETCH_SIZE" "$O/${P}_pmc_${w}_WRITE_SIZE" "$C" $cps > "$O/${P}_pmc_traffic_$w.json"
done
for c in FETCH_SIZE WRITE_SIZE; do pmc dsg $c --model dsgdetr --steps 3 --warmup 1; done
python3 tools/pmc_traffic.py "$O/${P}_pmc_dsg_FETCH_SIZE" "$O/${P}_pmc_dsg_WRITE_SIZE" "$C" 64 > "$O/${P}_pmc_traffic_dsgdetr_16x12.json"
# 3. MFMA-pipe occupancy of the dominant kernels (one counter per pass)
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU; do pmc busy $c --steps 3 --warmup 1; done
for k in "gemm16_kernel<sttran::Tile16<128, 176>" "gemm16_kernel<sttran::Tile16<128, 128>" "pair_conv_fused_kernel"; do
  tag=$(echo "$k" | tr -c 'A-Za-z0-9' '_' | cut -c1-40)
  python3 tools/pmc_mfma_busy.py "$O/${P}_pmc_busy_" "$k" "$C" > "$O/${P}_pmc_mfma_busy_$tag.json"
done
# ... and of every kernel class of the step side by side, the non-GEMM kernels included (round 5)
python3 tools/pmc_kernels.py "$O/${P}_pmc_busy_" "$C" mask_conv1_pool_kernel attention_short_kernel layernorm_kernel pair_conv_fused_kernel \
  "gemm16c_kernel<sttran::Tile16C<4>" "gemm16c_kernel<sttran::Tile16C<2>" "Tile16<128, 176>" "Tile16<128, 128>" > "$O/${P}_pmc_kernels.json"
# ... and of the second engine's kernels (round 6: gemm16x3_kernel / pair_conv_fused_x3_kernel / gemm16x3c_kernel / split_fm_kernel)
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU; do pmc busyx3 $c --gemm-engine bf16x3 --steps 3 --warmup 1; done
python3 tools/pmc_kernels.py "$O/${P}_pmc_busyx3_" "$C" "gemm16x3_kernel<sttran::Tile16<128, 176>" "gemm16x3_kernel<sttran::Tile16<128, 128>" \
  "pair_conv_fused_x3_kernel" "gemm16x3c_kernel" "split_fm_kernel" "gemm_x3_kernel" "layernorm_kernel" > "$O/${P}_x3_pmc.json"
# 4. bench lines (unprofiled)
# (stdout = the one compact line the driver parses; the full object goes to $BENCH_DETAIL)
BENCH_DETAIL="$O/${P}_bench_default_detail.json" python3 bench.py > "$O/${P}_bench_default_with_cpu.json" 2> "$O/${P}_bench_default.err"
# the driver's own command
BENCH_DETAIL="$O/${P}_bench_driver_cmd_detail.json" python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$O/${P}_bench_driver_cmd.json" 2>/dev/null
python3 bench.py --workload 64x36 --steps 20 --no-cpu-baseline --no-extra-workloads --detail "$O/${P}_bench_64x36_detail.json" > "$O/${P}_bench_64x36.json" 2>/dev/null
python3 bench.py --workload 64x36 --clips-per-step 1 --steps 20 --no-cpu-baseline --no-extra-workloads --detail "$O/${P}_bench_64x36_single_clip_detail.json" > "$O/${P}_bench_64x36_single_clip.json" 2>/dev/null
python3 bench.py --clips-per-step 1 --no-cpu-baseline --no-extra-workloads --detail "$O/${P}_bench_16x12_single_clip_detail.json" > "$O/${P}_bench_16x12_single_clip.json" 2>/dev/null
python3 bench.py --clips-per-step 16 --no-cpu-baseline --no-extra-workloads --detail "$O/${P}_bench_16x12_16clips_detail.json" > "$O/${P}_bench_16x12_16clips.json" 2>/dev/null
python3 bench.py --model dsgdetr --no-cpu-baseline --detail "$O/${P}_bench_dsgdetr_16x12_detail.json" > "$O/${P}_bench_dsgdetr_16x12.json" 2>/dev/null
python3 bench.py --model dsgdetr --workload 64x36 --steps 10 --no-cpu-baseline --detail "$O/${P}_bench_dsgdetr_64x36_detail.json" > "$O/${P}_bench_dsgdetr_64x36.json" 2>/dev/null
python3 tools/ag_split_bench.py > "$O/${P}_ag_split_shaped.json" 2>/dev/null
# RCCL itself on this one GPU: a one-rank nccl group under the gather / all-reduce / barrier code of the N > 1 legs
python3 bench.py --gpus 1 --rccl-selftest > "$O/${P}_rccl_selftest.json" 2> "$O/${P}_rccl_selftest.err" || true
# two ranks on this one GPU over gloo (the N > 1 code path, self-launched): what the 8-GPU driver run will execute
BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline \
  --detail "$O/${P}_bench_2ranks_gloo_one_gpu_detail.json" > "$O/${P}_bench_2ranks_gloo_one_gpu.json" 2> "$O/${P}_bench_2ranks.err" || true
# ... and eight ranks the same way: the dry run of the driver's one 8-GPU shot (record shape, gather, LPT, merged recall)
BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0 python3 bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --clips-per-step 8 \
  --strong-clips 16 --ag-clips 256 --detail "$O/${P}_bench_8ranks_gloo_one_gpu_detail.json" > "$O/${P}_bench_8ranks_gloo_one_gpu.json" 2> "$O/${P}_bench_8ranks.err" || true
# RCCL / gloo print banners on stdout ahead of the line: the committed files keep the line only
for f in rccl_selftest bench_2ranks_gloo_one_gpu bench_8ranks_gloo_one_gpu; do
  grep '^{' "$O/${P}_$f.json" > "$O/${P}_$f.json.tmp" || true
  mv "$O/${P}_$f.json.tmp" "$O/${P}_$f.json"
done
# the second engine alone on its shapes (kernel on pre-split operands, split pass, round 2's kernel, the exact engine)
python3 tools/x3_bench.py --shapes path16x64 > "$O/${P}_x3_bench.txt" 2>/dev/null || true
# raw rocprofv3 output is scratch: only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf "$O/${P}"_kt_*/ "$O/${P}"_pmc_*_[A-Z]*/
find "$O" -maxdepth 1 -name "${P}_pmc_*_[A-Z]*.err" -size -1k -delete
ls "$O" | grep "^${P}_" | head -80
du -sh "$O"
