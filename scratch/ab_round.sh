#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== twist"; timeout 120 scratch/bin/solve_twist 2>&1 | grep -v "block:\|band :\|ticks" | grep -v "^band"
echo "== one front"; DBA_SOLVE_TWIST=0 timeout 120 scratch/bin/solve_twist 2>&1 | grep -E "n=138|MISMATCH" -A2 | grep -v "block:\|band :\|ticks"
} > gpurun_out/solve_ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_ba.py -x -q 2>&1 | tail -3 > gpurun_out/tests.log
