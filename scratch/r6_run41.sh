#!/bin/bash
for rep in 1 2 3; do
for a in cur l0split; do
  lib=$PWD/scratch/abl/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  echo -n "$a  "; DBA_HIP_LIB=$lib python scratch/motion_prof.py 2>&1 | grep "motion filter"
done; done
DBA_HIP_LIB=$PWD/dba-fusion_amd/lib/libdba_hip.so python scratch/build_ab.py cur 2>&1 | grep "64x64\|48x64"
