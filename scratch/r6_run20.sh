#!/bin/bash
for rep in 1 2; do
for a in base abl_mfma abl_l0 abl_pool abl_nostore abl_mfmal0 abl_skel; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = base ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|55x55"
done; done
