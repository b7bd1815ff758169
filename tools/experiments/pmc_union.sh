# MFMA-busy / wait counters of the union conv and the conv3x3 next to the nn.Linear kernel (one counter per pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/u_pmcm_$c -- python3 bench.py --steps 3 --warmup 1 --profile-only-batch > /dev/null 2> gpurun_out/u_pmcm_$c.err
done
for k in EpiUnionFlat EpiConvRelBn EpiLinearV; do python3 tools/pmc_mfma_busy.py gpurun_out/u_pmcm_ $k x > gpurun_out/u_$k.json; cat gpurun_out/u_$k.json; done
