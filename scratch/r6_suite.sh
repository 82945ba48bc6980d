#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export DBA_PARITY_REPORT=$OUT/r06_parity_report.jsonl
rm -f $DBA_PARITY_REPORT
( time timeout 2400 python -m pytest tests -q -m gpu ) > $OUT/r6_pytest_gpu.txt 2>&1; tail -8 $OUT/r6_pytest_gpu.txt
