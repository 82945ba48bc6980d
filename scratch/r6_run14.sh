#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
for w in 25_96 64_512; do echo "== $w"; DBA_HIP_LIB=$PWD/scratch/libdba_hip_linprof.so python scratch/lin_prof.py $w 2>&1 | tail -4; done > $OUT/r6_lin_prof3.txt 2>&1
cat $OUT/r6_lin_prof3.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r6_trace_y
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6_trace_y -- python /root/repo/bench.py --window 64_512 --steps 10 --warmup 4 --no-cpu-baseline --no-extras > $OUT/r6_trace_y.log 2>&1
f=$(ls -t $(find $OUT/r6_trace_y -name "*kernel_trace.csv") | head -1)
head -1 $f | cut -d, -f10- ; grep ba_linearize $f | tail -2 | cut -d, -f10-; grep schur_gram $f | tail -1 | cut -d, -f10-
