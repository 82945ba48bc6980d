#!/bin/bash
# builds a variant of the library: scratch/abl/libdba_hip_<TAG>.so with -D<flags> applied to ONE source
#   tools/build_variant.sh <tag> <source stem, e.g. corr_sheared> [-DX=1 ...]
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift; shift
mkdir -p scratch/abl build/abl_$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude"
EXACT=""
case $SRC in corr_lookup|corr_sheared|altcorr) EXACT="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc $FLAGS $EXACT "$@" -c dba-fusion_amd/csrc/$SRC.hip -o build/abl_$TAG/$SRC.o
OBJS=$(ls build/gfx950/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/abl/libdba_hip_$TAG.so $OBJS build/abl_$TAG/$SRC.o
echo built scratch/abl/libdba_hip_$TAG.so
