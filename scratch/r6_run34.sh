#!/bin/bash
OUT=$PWD/gpurun_out
for rep in 1 2 3; do
for a in head cur; do
  lib=$PWD/scratch/abl/libdba_hip_$a.so; [ $a = cur ] && lib=$PWD/dba-fusion_amd/lib/libdba_hip.so
  DBA_HIP_LIB=$lib python scratch/build_ab.py $a 2>&1 | grep "us/edge"
done; done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $OUT/r6_pytest_gpu_e.txt; cat $OUT/r6_pytest_gpu_e.txt
