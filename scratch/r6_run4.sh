#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_solve_cold.py -x -q -m gpu ) > $OUT/r6_cold_test.txt 2>&1; tail -8 $OUT/r6_cold_test.txt
( echo "== shipped library"; python tests/test_gpu_solve_cold.py dba-fusion_amd/lib/libdba_hip.so 60
  echo "== round 5's hand-over race compiled back in (scratch/build_unfixed_lib.sh)"; python tests/test_gpu_solve_cold.py scratch/libdba_hip_unfixed.so 60 ) > $OUT/r6_cold_start_stress.txt 2>&1
cat $OUT/r6_cold_start_stress.txt | cut -c1-200 | head -40
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r6_pytest_gpu_a.txt 2>&1; tail -5 $OUT/r6_pytest_gpu_a.txt
