#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export DBA_PARITY_REPORT=$OUT/r6_parity_report.jsonl
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_compiled_adapter.py -q -m gpu > $OUT/r6_pytest_gpu_b.txt 2>&1; tail -8 $OUT/r6_pytest_gpu_b.txt
