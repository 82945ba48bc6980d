#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 64_512; do
rm -rf $OUT/r6_trace_x
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6_trace_x -- python $REPO/bench.py --window $w --steps 30 --warmup 6 --no-cpu-baseline --no-extras > $OUT/r6_trace_x.log 2>&1
cp $(ls -t $(find $OUT/r6_trace_x -name "*kernel_stats.csv") | head -1) $OUT/r6_kernel_stats_${w}_b.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/r6_kernel_stats_${w}_b.csv")))[:12]:
    if 'dba' in r['Name']: print("$w", r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'])
PY
done
rm -rf $OUT/r6_trace_x
