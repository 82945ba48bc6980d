#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r6_trace_64
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6_trace_64 -- python $REPO/bench.py --window 64_512 --steps 30 --warmup 6 --no-cpu-baseline --no-extras > $OUT/r6_trace_64.log 2>&1
cp $(ls -t $(find $OUT/r6_trace_64 -name "*kernel_stats.csv") | head -1) $OUT/r6_kernel_stats_64_512_a.csv
head -12 $OUT/r6_kernel_stats_64_512_a.csv | cut -c1-150
rm -rf $OUT/r6_trace_64
