#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== default"; timeout 200 python scratch/lookup_py_bench.py 2>&1 | tail -4 | head -3
for v in d3 d4 w4 o6 o5d3 b3 b6; do echo "== $v"; DBA_HIP_LIB=/root/repo/scratch/abl/libdba_hip_$v.so timeout 200 python scratch/lookup_py_bench.py 2>&1 | tail -4 | head -3; done
echo "== default again"; timeout 200 python scratch/lookup_py_bench.py 2>&1 | tail -4 | head -3
} > gpurun_out/lookup_cfg.log 2>&1
