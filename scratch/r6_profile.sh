#!/bin/bash
# round 6: the round's profile set (tools/profile_all.sh r06) + the GPU suite with its parity report
OUT=$PWD/gpurun_out; mkdir -p $OUT
export DBA_PARITY_REPORT=$OUT/r06_parity_report.jsonl
rm -f $DBA_PARITY_REPORT
( time timeout 2400 python -m pytest tests -q -m gpu ) > $OUT/r06_pytest_gpu.txt 2>&1; tail -4 $OUT/r06_pytest_gpu.txt
bash tools/profile_all.sh r06 > $OUT/r06_profile_all.log 2>&1
tail -5 $OUT/r06_profile_all.log
