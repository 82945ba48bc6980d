#!/bin/bash
# round 6: the round's profile set (tools/profile_all.sh r06) + the parity report of the GPU suite's BA tests
OUT=$PWD/gpurun_out; mkdir -p $OUT
export DBA_PARITY_REPORT=$OUT/r06_parity_report.jsonl
rm -f $DBA_PARITY_REPORT
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_reference_caller.py tests/test_gpu_caller_sequence.py -q -m gpu > $OUT/r06_pytest_ba.txt 2>&1; tail -3 $OUT/r06_pytest_ba.txt
bash tools/profile_all.sh r06 > $OUT/r06_profile_all.log 2>&1
tail -30 $OUT/r06_profile_all.log
