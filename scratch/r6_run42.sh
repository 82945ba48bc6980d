#!/bin/bash
for rep in 1 2 3; do python scratch/motion_prof.py 2>&1 | grep "motion filter"; done
timeout 900 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_shapes.py tests/test_gpu_corr_slots.py tests/test_gpu_reference_caller.py tests/test_gpu_caller_sequence.py tests/test_gpu_soak.py tests/test_gpu_entrypoints.py -x -q -m gpu 2>&1 | tail -3
