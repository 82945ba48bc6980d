#!/bin/bash
for rep in 1 2; do
for a in base ab2_l0oor ab2_l0nolds ab2_l0 ab2_pooloff ab2_pooloff_l0oor; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = base ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64"
done; done
