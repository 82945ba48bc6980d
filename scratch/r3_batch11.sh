#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b11; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_solve.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -5
run() {
  local label=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --window $w --steps 60 --warmup 12 --no-cpu-baseline --no-extras 2>$O/err_${label}_$w.txt | python -c "
import json,sys
try:
    d=json.load(sys.stdin); print('$label $w', d['value'], 'upd/s  ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_ms'], 'ba', d['extra']['ba_itrs2_us_p50'])
except Exception as e: print('$label $w FAILED', e)"
}
run plan 64_512 X=1
run noplan 64_512 DBA_WS_CACHE=0
run plan 32_122 X=1
run noplan 32_122 DBA_WS_CACHE=0
run weak1 25_96 X=1
timeout 300 python bench.py --scaling weak --steps 60 --warmup 12 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('weak N=1', d['value'], d['ms_per_step'], d['extra']['edges_per_s'], d['extra']['ba_itrs2_us_p50'])"
