#!/bin/bash
# tile-order group width of the bf16x3 kernel (N-tiles of one M-tile that are consecutive in the tile order): experiment build
LIB=${1:-tools/experiments/prev/csrc_exp/libsttran_hip.so}
for sh in 21120,1936,1936 21120,5808,1936 21120,2048,1936; do
  for g in 8 4 11 16 2 8; do
    echo -n "group_n=$g "
    STTRAN_LIB=$LIB STTRAN_X3_GROUP_N=$g python tools/x3_bench.py --one $sh --iters 8 2>&1 | grep -v amdgpu.ids | cut -c1-110
  done
done
