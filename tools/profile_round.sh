set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=538c50d
# 1. kernel traces (per-step averages: --profile-only-batch)
for w in 16x12 64x36; do
  st=20; [ $w = 64x36 ] && st=10
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e_kt_$w -- python3 bench.py --workload $w --steps $st --warmup 3 --profile-only-batch > gpurun_out/r2e_kt_$w.json 2> gpurun_out/r2e_kt_$w.err
done
# 2. single clip trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e_kt_single -- python3 bench.py --clips-per-step 1 --steps 50 --warmup 3 --profile-only-batch > gpurun_out/r2e_kt_single.json 2> gpurun_out/r2e_kt_single.err
# 3. PMC traffic passes (separate passes, kernel-trace only)
for w in 16x12 64x36; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r2e_pmc_${w}_$c -- python3 bench.py --workload $w --steps 3 --warmup 1 --profile-only-batch > /dev/null 2> gpurun_out/r2e_pmc_${w}_$c.err
  done
  cps=64; [ $w = 64x36 ] && cps=4
  python3 tools/pmc_traffic.py gpurun_out/r2e_pmc_${w}_FETCH_SIZE gpurun_out/r2e_pmc_${w}_WRITE_SIZE $C $cps > gpurun_out/r2e_pmc_traffic_$w.json
done
# 4. MFMA busy passes
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r2e_pmcm_$c -- python3 bench.py --steps 3 --warmup 1 --profile-only-batch > /dev/null 2> gpurun_out/r2e_pmcm_$c.err
done
python3 tools/pmc_mfma_busy.py gpurun_out/r2e_pmcm_ "GemmTile<256, 128, 4, 2, 3>, sttran::EpiLinearV" $C > gpurun_out/r2e_pmc_mfma_busy.json
# 5. bench lines (unprofiled)
python3 bench.py > gpurun_out/r2e_bench_default_with_cpu.json 2> gpurun_out/r2e_bench_default.err
python3 bench.py --workload 64x36 --steps 20 --no-cpu-baseline --no-extra-workloads > gpurun_out/r2e_bench_64x36.json 2>/dev/null
python3 bench.py --workload 64x36 --clips-per-step 1 --steps 20 --no-cpu-baseline --no-extra-workloads > gpurun_out/r2e_bench_64x36_single_clip.json 2>/dev/null
python3 bench.py --clips-per-step 1 --no-cpu-baseline --no-extra-workloads > gpurun_out/r2e_bench_16x12_single_clip.json 2>/dev/null
python3 bench.py --clips-per-step 16 --no-cpu-baseline --no-extra-workloads > gpurun_out/r2e_bench_16x12_16clips.json 2>/dev/null
python3 bench.py --model dsgdetr --no-cpu-baseline > gpurun_out/r2e_bench_dsgdetr_16x12.json 2>/dev/null
python3 bench.py --model dsgdetr --workload 64x36 --steps 10 --no-cpu-baseline > gpurun_out/r2e_bench_dsgdetr_64x36.json 2>/dev/null
python3 tools/ag_split_bench.py > gpurun_out/r2e_ag_split_shaped.json 2>/dev/null
ls gpurun_out/r2e_kt_16x12/*/ | head; find gpurun_out -name "*kernel_stats.csv" | head
du -sh gpurun_out
