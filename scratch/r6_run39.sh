#!/bin/bash
OUT=$PWD/gpurun_out
for rep in 1 2 3; do
  python scratch/motion_prof.py 2>&1 | grep "motion filter"
  python scratch/build_ab.py cur 2>&1 | grep "64x64\|48x64"
done
timeout 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_shapes.py tests/test_gpu_corr_slots.py tests/test_gpu_reference_caller.py tests/test_gpu_caller_sequence.py -x -q -m gpu 2>&1 | tail -3
