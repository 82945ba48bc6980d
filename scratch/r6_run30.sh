#!/bin/bash
for rep in 1 2; do
for a in cur x_mfma x_l0oor x_pooloor x_alloor x_skel; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|48x64"
done; done
