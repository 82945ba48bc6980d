#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== measured skyline"; timeout 120 scratch/bin/solve_twist 2>&1 | grep -E "MISMATCH|tile :|stages" | grep -v "0.[0-9]* us per\|0.00 factor" | head -4
echo "== graph skyline"; HARNESS_FPOSE=1 timeout 120 scratch/bin/solve_twist 2>&1 | grep -v "block:\|band :\|ticks" | grep -v "^band"
} > gpurun_out/solve_ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_ba.py -x -q 2>&1 | tail -3 > gpurun_out/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_sky.json 2>/dev/null
