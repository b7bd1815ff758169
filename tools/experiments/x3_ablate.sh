#!/bin/bash
# Where does the bf16x3 kernel's time go: timing-only ablations of gemm16x3_kernel (wrong results) on an EXPERIMENT build
# (make EXTRA=-DSTTRAN_GEMM_EXPERIMENT; STTRAN_LIB points at it).  1 = no epilogue stores, 2 = no barrier in the K loop,
# 3 = no LDS-DMA in the loop, 4 = no A-fragment loads in the loop, 5 = 3 + 4, 6 = all loads, but always L2 hits.
LIB=${1:-tools/experiments/prev/csrc_exp/libsttran_hip.so}
for sh in 21120,1936,1936 21120,5808,1936; do
  for a in ${ABLS:-0 1 2 3 4 5 6 0}; do
    echo -n "ablate=$a "
    STTRAN_LIB=$LIB STTRAN_X3_ABLATE=$a python tools/x3_bench.py --one $sh --iters 8 2>&1 | grep -v amdgpu.ids | cut -c1-110
  done
done
