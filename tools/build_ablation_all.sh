#!/bin/bash
# full rebuild of the library with extra -D flags into scratch/abl/libdba_hip_<TAG>.so (parameter sweeps)
set -e
cd /root/repo
TAG=$1; shift
mkdir -p scratch/abl build/all_$TAG
for f in dba-fusion_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  case $b in corr_lookup|corr_sheared|altcorr) extra="-ffp-contract=off";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude $extra "$@" -c $f -o build/all_$TAG/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/abl/libdba_hip_$TAG.so build/all_$TAG/*.o
