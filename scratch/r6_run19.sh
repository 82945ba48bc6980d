#!/bin/bash
for rep in 1 2; do
for a in base aux1 aux2 aux3 aux16 aux17 aux18 aux19; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = base ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep -v amdgpu.ids
done; done
