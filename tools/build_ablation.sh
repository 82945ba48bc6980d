#!/bin/bash
# builds ablation variants of the library: scratch/abl/libdba_hip_<TAG>.so with -D<flags> applied to corr_sheared.hip
set -e
cd /root/repo
TAG=$1; shift
mkdir -p scratch/abl build/abl_$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude"
hipcc $FLAGS -ffp-contract=off "$@" -c dba-fusion_amd/csrc/corr_sheared.hip -o build/abl_$TAG/corr_sheared.o
OBJS=$(ls build/gfx950/*.o | grep -v corr_sheared.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/abl/libdba_hip_$TAG.so $OBJS build/abl_$TAG/corr_sheared.o
