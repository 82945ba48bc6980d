#!/bin/bash
OUT=$PWD/gpurun_out
for rep in 1 2 3; do
for a in pool8 cur; do
  lib=$PWD/scratch/abl/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "64x64\|48x64"
done; done
for a in f16prof7; do
  DBA_HIP_LIB=$PWD/scratch/abl/libdba_hip_$a.so python scratch/build_ab.py $a 2>&1 | grep "F16_PROF n=32" > $OUT/r6_f16prof_$a.txt; head -2 $OUT/r6_f16prof_$a.txt
done
timeout 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_shapes.py tests/test_gpu_corr_slots.py tests/test_gpu_reference_caller.py -x -q -m gpu 2>&1 | tail -3
