#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b5; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
run() {
  local label=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --window $w --steps 60 --warmup 12 --no-cpu-baseline --no-extras 2>$O/err_${label}_$w.txt | python -c "
import json,sys
try:
    d=json.load(sys.stdin); print('$label $w', d['value'], 'upd/s  lookup', d['roofline']['avg_launch_ms'], 'ba', d['extra']['ba_itrs2_us_p50'])
except Exception as e: print('$label $w FAILED', e)"
}
for w in 25_96 64_512 32_122; do
  run default $w X=1
  run nofuse $w DBA_BA_FUSE_UPDATE=0
  run nocache $w DBA_WS_CACHE=0
  run neither $w DBA_WS_CACHE=0 DBA_BA_FUSE_UPDATE=0
done
cd /tmp && export TMPDIR=/tmp
for w in 25_96 64_512; do
  rm -rf $O/trace_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $REPO/bench.py --window $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/trace_$w.log 2>&1
  echo "== kernel stats $w"; python $REPO/tools/kstats.py $O/trace_$w | grep -v "Cat\|copyBuffer\|gather\|pixel_major\|build_fused"
done
find $O -name "*.csv" -size +2M -delete
