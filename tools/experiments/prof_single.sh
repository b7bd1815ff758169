cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3s_kt_single
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3s_kt_single -- python3 bench.py --clips-per-step 1 --steps 50 --warmup 3 --profile-only-batch > gpurun_out/r3s_kt_single.json 2> gpurun_out/r3s_kt_single.err
python3 - <<'PY'
import glob, pandas as pd
f=glob.glob('gpurun_out/r3s_kt_single/*/*kernel_stats.csv')[0]
t=pd.read_csv(f)
t['per_step_us']=t['TotalDurationNs']/1e3/53
t['calls_per_step']=t['Calls']/53
pd.set_option('display.width',250); pd.set_option('display.max_colwidth',110)
print(t[['Name','calls_per_step','AverageNs','per_step_us']].sort_values('per_step_us',ascending=False).head(25).to_string())
print('total per step us', t['per_step_us'].sum())
PY
