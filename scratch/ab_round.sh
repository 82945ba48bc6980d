#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_corr.py tests/test_gpu_entrypoints.py -x -q 2>&1 | tail -3 > gpurun_out/tests.log
