#!/bin/bash
# Where do the two 16x16x4 convolution kernels lose their time?  Experiment build of the library (make EXTRA=-DSTTRAN_GEMM_EXPERIMENT,
# given as $1), STTRAN_T16C_ABLATE = 0..7 (gemm_f32_t16c.h: timing-only variants, wrong results), mean microseconds of the two
# kernels in situ from bench.py's per-kernel table (HIP events around every launch).
LIB=${1:?path of an experiment build of libsttran_hip.so}
for a in ${ABLATIONS:-0 1 2 3 4 5 6 7 8 0}; do
  STTRAN_LIB=$LIB STTRAN_T16C_ABLATE=$a python3 bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-extra-workloads 2>&1 >/dev/null \
    | grep "^BENCH_DETAIL" | cut -c14- | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
row={k['kernel'][:34]: (round(k['mean_us']), round(k.get('tflops',0),1)) for k in r['by_kernel'] if '16c' in k['kernel']}
print('ablate $a', row)"
done
