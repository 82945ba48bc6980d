#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== twist"; timeout 120 scratch/bin/solve_twist 2>&1 | grep -v "block:\|band :\|ticks" | grep -v "^band"
} > gpurun_out/solve_ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_ba.py -x -q 2>&1 | tail -5 > gpurun_out/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_dual.json 2> gpurun_out/bench_dual.err
