#!/bin/bash
# experiment only: the library with in-kernel timestamps in the linearisation (-DLIN_PROF) -> scratch/libdba_hip_linprof.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build/linprof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -DLIN_PROF ${EXTRA_DEFS} \
    -c dba-fusion_amd/csrc/ba_kernels.hip -o build/linprof/ba_kernels.o
objs=$(ls build/gfx950/*.o | grep -v ba_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libdba_hip_linprof.so $objs build/linprof/ba_kernels.o
