#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
for lib in linprof linhalf; do for w in 25_96 64_512; do echo "== $lib $w"; DBA_HIP_LIB=$PWD/scratch/libdba_hip_$lib.so python scratch/lin_prof.py $w 2>&1 | tail -1; done; done > $OUT/r6_lin_prof2.txt 2>&1
cat $OUT/r6_lin_prof2.txt
