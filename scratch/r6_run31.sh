#!/bin/bash
OUT=$PWD/gpurun_out
for rep in 1 2 3; do
for a in base16 cur; do
  lib=$PWD/scratch/abl/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|48x64"
done; done
for a in f16prof f16prof2; do
  DBA_HIP_LIB=$PWD/scratch/abl/libdba_hip_$a.so python scratch/build_ab.py $a 2>&1 | grep "F16_PROF n=32" > $OUT/r6_f16prof_$a.txt; head -6 $OUT/r6_f16prof_$a.txt; tail -4 $OUT/r6_f16prof_$a.txt
done
