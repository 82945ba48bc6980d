#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b4; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python bench.py --steps 40 --warmup 8 > $O/bench_25_96.json 2> $O/bench_25_96.err; cat $O/bench_25_96.json | cut -c1-3000
timeout 300 python bench.py --steps 40 --warmup 8 --scaling weak --no-extras > $O/bench_weak1.json 2> $O/bench_weak1.err; cat $O/bench_weak1.json | cut -c1-1500
