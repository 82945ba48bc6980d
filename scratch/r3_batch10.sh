#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b10; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_sharded.py tests/test_gpu_caller_sequence.py -m gpu -q 2>&1 | tail -5
run() {
  local label=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --window $w --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>$O/err_${label}_$w.txt | python -c "
import json,sys
try:
    d=json.load(sys.stdin); print('$label $w', d['value'], 'upd/s  lookup', d['roofline']['avg_launch_ms'], 'ba', d['extra']['ba_itrs2_us_p50'])
except Exception as e: print('$label $w FAILED', e)"
}
run w8nch4 64_512 X=1
run w4nch4 64_512 DBA_SCHUR_WAVES=4
run w8nch8 64_512 DBA_SCHUR_NCH=8
run w8nch2 64_512 DBA_SCHUR_NCH=2
run w4nch8 64_512 DBA_SCHUR_WAVES=4 DBA_SCHUR_NCH=8
run w8 25_96 DBA_SCHUR_KERNEL=frame
run w4 25_96 DBA_SCHUR_KERNEL=frame DBA_SCHUR_WAVES=4
run w8nch8 25_96 DBA_SCHUR_KERNEL=frame DBA_SCHUR_NCH=8
run rows 25_96 X=1
