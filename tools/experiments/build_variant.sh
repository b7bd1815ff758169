#!/bin/bash
# build_variant.sh <name> <source.hip> <extra flags...>: ONE translation unit recompiled with extra flags, linked with the tree's
# other objects into nl-vsgg_amd/csrc/ab/libsttran_hip_<name>.so (git-ignored; travels with the gpurun snapshot).  Select it with
# STTRAN_LIB=<path> for same-box A/B runs.  Run `make -C nl-vsgg_amd/csrc` first (the other objects must be current).
set -euo pipefail
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../../nl-vsgg_amd/csrc"
mkdir -p ab
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -c "$src" -o "ab/${src%.hip}_$name.o"
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "ab/libsttran_hip_$name.so" $objs "ab/${src%.hip}_$name.o"
rm -f "ab/${src%.hip}_$name.o"
echo "built nl-vsgg_amd/csrc/ab/libsttran_hip_$name.so"
