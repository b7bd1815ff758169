#!/bin/bash
# product library + a trace build of the mask-conv kernel (s_memtime stamps, tools/experiments/mc_trace.py) into csrc/ab/
set -e
cd /root/repo/nl-vsgg_amd/csrc
make -j8 2>&1 | grep -E "error|Error" || true
mkdir -p ab /tmp/mcasm
/opt/rocm/bin/hipcc -DSTTRAN_GEMM_EXPERIMENT -DSTTRAN_MC_TRACE -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function \
  -c kernels_maskconv.hip -o /tmp/mcasm/kernels_maskconv_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libsttran_hip_mctrace.so $(ls *.o | grep -v kernels_maskconv.o) /tmp/mcasm/kernels_maskconv_trace.o
ls -la libsttran_hip.so ab/libsttran_hip_mctrace.so
