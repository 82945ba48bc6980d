#!/bin/bash
for rep in 1 2; do
for a in new ab4_geom ab4_poolcomp ab4_geompool; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = new ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64"
done; done
