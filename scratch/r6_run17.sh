#!/bin/bash
OUT=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ba.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
for lib in scratch/libdba_hip_base.so dba-fusion_amd/lib/libdba_hip.so; do
for w in 25_96 64_512; do
DBA_HIP_LIB=$PWD/$lib python bench.py --window $w --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', d['value'], d['ms_per_step'], 'ba', d['extra']['ba_itrs2_us_p50'])"
done; done; done
