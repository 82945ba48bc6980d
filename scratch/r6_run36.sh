#!/bin/bash
OUT=$PWD/gpurun_out
for rep in 1 2 3; do
for a in cur w16 prio2 w16prio; do
  lib=$PWD/scratch/abl/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|48x64"
done; done
