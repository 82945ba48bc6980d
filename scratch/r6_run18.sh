#!/bin/bash
OUT=$PWD/gpurun_out
for i in 1 2; do python bench.py --scaling weak --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], 'look', d['extra'].get('lookup_us_p10_p50_p90'), 'ba', d['extra'].get('ba_itrs2_us_p50'), d['config'].get('edges'))"; done
DBA_SOLVE_FRONTS=0 python bench.py --scaling weak --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one front:', d['value'], d['ms_per_step'], 'ba', d['extra'].get('ba_itrs2_us_p50'))"
