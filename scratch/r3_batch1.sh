#!/bin/bash
# round 3, GPU batch 1: per-source-frame Schur kernel (tests, A/B against the row-pair kernel), memory-structure microbenchmark
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for k in gram rows; do
  for w in 25_96 64_512 32_122; do
    DBA_SCHUR_KERNEL=$k timeout 300 python bench.py --window $w --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_${k}_$w.json 2> $O/bench_${k}_$w.err
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_${k}_$w.json"))
    print("$k $w", d["value"], "upd/s", d["ms_per_step"], "ms  lookup", d["roofline"]["avg_launch_ms"], "ba", d["extra"]["ba_itrs2_us_p50"])
except Exception as e:
    print("$k $w FAILED", e)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for w in 25_96 64_512; do
  rm -rf $O/trace_$w
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python $REPO/bench.py --window $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/trace_$w.log 2>&1
  echo "== kernel stats $w"; python $REPO/tools/kstats.py $O/trace_$w | sort -k6 -n -r | head -14
done
for n in 2 8; do
  DBA_SCHUR_NCH=$n timeout 300 python $REPO/bench.py --window 25_96 --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('nch=$n 25_96', d['value'], d['extra']['ba_itrs2_us_p50'])"
  DBA_SCHUR_NCH=$n timeout 300 python $REPO/bench.py --window 64_512 --steps 40 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('nch=$n 64_512', d['value'], d['extra']['ba_itrs2_us_p50'])"
done
cd $REPO
echo "== mem_pattern2 96"; timeout 120 scratch/bin/mem_pattern2 96 | tee $O/mem_pattern2_96.txt
echo "== mem_pattern2 384"; timeout 120 scratch/bin/mem_pattern2 384 | tee $O/mem_pattern2_384.txt
echo "== mem_pattern2 96 pad 128"; timeout 120 scratch/bin/mem_pattern2 96 128 | tee $O/mem_pattern2_96_pad128.txt
find $O -name "*.csv" -size +2M -delete
