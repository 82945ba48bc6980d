#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
( echo "== round 5's library before the fix (git 26f49d0^, built from that tree)"; python tests/test_gpu_solve_cold.py scratch/libdba_hip_26f49d0_parent.so 80 ) > $OUT/r6_cold_start_stress_old.txt 2>&1
cut -c1-220 $OUT/r6_cold_start_stress_old.txt | head -30
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r6_pytest_gpu_a.txt 2>&1; tail -5 $OUT/r6_pytest_gpu_a.txt
