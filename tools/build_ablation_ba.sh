#!/bin/bash
# ablation variants of the library with -D<flags> applied to ba_kernels.hip: scratch/abl/libdba_hip_<TAG>.so
set -e
cd /root/repo
TAG=$1; shift
mkdir -p scratch/abl build/abl_$TAG
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude "$@" -c dba-fusion_amd/csrc/ba_kernels.hip -o build/abl_$TAG/ba_kernels.o
OBJS=$(ls build/gfx950/*.o | grep -v ba_kernels.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/abl/libdba_hip_$TAG.so $OBJS build/abl_$TAG/ba_kernels.o
