#!/bin/bash
# mask_conv1_pool_kernel under a start-up stagger of the second workgroup of every CU (STTRAN_MC_STAGGER = rounds of s_sleep 127,
# experiment build given as $1): in-situ mean microseconds from bench.py's per-kernel table.
LIB=${1:?path of an experiment build of libsttran_hip.so}
for a in ${STAGGERS:-0 1 2 3 4 5 6 8 0}; do
  STTRAN_LIB=$LIB STTRAN_MC_STAGGER=$a python3 bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-extra-workloads 2>&1 >/dev/null \
    | grep "^BENCH_DETAIL" | cut -c14- | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
row={k['kernel'][:34]: (round(k['mean_us'],1), round(k.get('tflops',0),1)) for k in r['by_kernel'] if k['class']=='mask_conv'}
print('stagger $a', row, 'value', round(d['value']))"
done
