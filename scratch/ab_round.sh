#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_solve.py tests/test_gpu_ba.py -x -q 2>&1 | tail -5 > gpurun_out/tests.log
