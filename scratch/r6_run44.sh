#!/bin/bash
for rep in 1 2; do
DBA_BUILD_WG_TARGET=256 python scratch/build_n.py old 2>&1 | grep "x"
python scratch/build_n.py new 2>&1 | grep "x"
done
python scratch/build_ab.py new 2>&1 | grep "us/edge"
timeout 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_shapes.py tests/test_gpu_corr_slots.py -x -q -m gpu 2>&1 | tail -2
