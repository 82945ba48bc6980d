#!/bin/bash
for rep in 1 2; do
for a in new ab3_oor ab3_nol1 ab3_nol23 ab3_none; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = new ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64"
done; done
