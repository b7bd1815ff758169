#!/bin/bash
# same-box A/B of builds of the 16x16x32 bf16x3 kernel (tools/experiments/build_variant.sh): x3_bench per shape and library, twice
VARS=${VARS:-"base ppb3 ppb4 ppb5"}
for rep in 1 2; do
for sh in 21120,1936,1936 21120,5808,1936 21120,2048,1936 11264,1936,2048; do
  for v in $VARS; do
    L=""; [ "$v" != base ] && L=nl-vsgg_amd/csrc/ab/libsttran_hip_$v.so
    echo -n "$v "
    STTRAN_LIB=$L python tools/x3_bench.py --one $sh --iters 10 2>&1 | grep -v amdgpu.ids | cut -c1-118
  done
done
done
