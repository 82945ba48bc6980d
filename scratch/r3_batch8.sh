#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b8; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 400 python bench.py --steps 40 --warmup 8 > $O/bench_25_96.json 2> $O/bench_25_96.err; python -c "
import json; d=json.load(open('$O/bench_25_96.json')); print(d['value'], d['roofline']['frac'], json.dumps(d['extra'])[:1500])"
