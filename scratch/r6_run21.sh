#!/bin/bash
for rep in 1 2; do
for a in base st_g2t200 st_g2t400 st_g2t600 st_g4t100 st_g4t200 st_g4t300 st_g8t100; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = base ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|55x55\|48x64"
done; done
