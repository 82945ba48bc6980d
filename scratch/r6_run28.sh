#!/bin/bash
for rep in 1 2; do
for a in cur v_l0low1 v_l0low0 v_l0low3; do
  lib=$PWD/scratch/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|48x64"
done
for cap in 8 32 64; do DBA_BUILD_SPW_CAP=$cap python scratch/build_ab.py cap$cap 2>&1 | grep "64x64\|48x64"; done
for wt in 128 512 1024; do DBA_BUILD_WG_TARGET=$wt python scratch/build_ab.py wgt$wt 2>&1 | grep "64x64\|48x64"; done
done
