#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -x 2>&1 | tail -8
