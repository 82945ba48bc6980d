#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b7; rm -rf $O; mkdir -p $O
ab() { # tag lib persist
  DBA_HIP_LIB=$2 DBA_LOOKUP_PERSIST=$3 timeout 300 python scratch/lookup_ab.py "$1" 25_96 2>&1 | grep -v "^\[W\|amdgpu.ids" | tail -2
}
ab base-oneshot "" 0
ab occ8-persist256 "" 256
ab occ8-persist512 "" 512
ab occ7-oneshot $REPO/scratch/abl/libdba_hip_occ7.so 0
ab occ7-persist256 $REPO/scratch/abl/libdba_hip_occ7.so 256
ab occ6-oneshot $REPO/scratch/abl/libdba_hip_occ6.so 0
ab occ6-persist192 $REPO/scratch/abl/libdba_hip_occ6.so 192
ab occ6-persist256 $REPO/scratch/abl/libdba_hip_occ6.so 256
