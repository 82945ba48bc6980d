#!/bin/bash
# round 6: the ring solver's harness, the solver tests, the 64/512 window (bench + kernel trace)
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
timeout 300 ./scratch/solve_wave_test_bin > $OUT/r6_harness.txt 2>&1; echo "harness rc $?" >> $OUT/r6_harness.txt
timeout 900 python -m pytest tests/test_gpu_solve.py -x -q -m gpu > $OUT/r6_test_solve.txt 2>&1; tail -5 $OUT/r6_test_solve.txt
timeout 600 python bench.py --window 64_512 --steps 40 --warmup 10 --no-cpu-baseline > $OUT/r6_bench_64_512_a.json 2> $OUT/r6_bench_64_512_a.err
python - <<PY
import json
d=json.loads(open("$OUT/r6_bench_64_512_a.json").read().strip().splitlines()[-1])
print("64/512: value", d["value"], "ms", d["ms_per_step"], "ba_itrs2_us_p50", d["extra"]["ba_itrs2_us_p50"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r6_trace_64
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6_trace_64 -- python $REPO/bench.py --window 64_512 --steps 30 --warmup 6 --no-cpu-baseline --no-extras > $OUT/r6_trace_64.log 2>&1
cp $(ls -t $(find $OUT/r6_trace_64 -name "*kernel_stats.csv") | head -1) $OUT/r6_kernel_stats_64_512_a.csv
head -8 $OUT/r6_kernel_stats_64_512_a.csv | cut -c1-150
rm -rf $OUT/r6_trace_64
cd $REPO
grep -c MISMATCH $OUT/r6_harness.txt; grep -A1 "P=63\|P=64\|P=60\|P=37\|P=38\|w=8\|w=10" $OUT/r6_harness.txt | grep -v "^--" | head -70
