#!/bin/bash
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r6_trace_m
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6_trace_m -- python /root/repo/scratch/motion_prof.py 2>&1 | grep "motion filter"
f=$(ls -t $(find $OUT/r6_trace_m -name "*kernel_stats.csv") | head -1)
cut -d, -f1-4 $f | cut -c1-160 | head -12
