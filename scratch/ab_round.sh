#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== twist"; timeout 120 scratch/bin/solve_twist 2>&1 | grep -E "MISMATCH|tile :|stages" | grep -v "0.3. us\|0.00 factor" | head -8
} > gpurun_out/solve_ab.log 2>&1
timeout 600 python -m pytest tests/test_gpu_solve.py tests/test_gpu_ba.py tests/test_gpu_corr.py -x -q 2>&1 | tail -5 > gpurun_out/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_twist.json 2> gpurun_out/bench_twist.err
DBA_SOLVE_TWIST=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_notwist.json 2>> gpurun_out/bench_twist.err
