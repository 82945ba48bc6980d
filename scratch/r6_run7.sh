#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export DBA_PARITY_REPORT=$OUT/r6_parity_report.jsonl
timeout 900 python -m pytest tests/test_gpu_fusion_export.py tests/test_gpu_ba.py tests/test_gpu_caller_sequence.py tests/test_gpu_compiled_adapter.py tests/test_gpu_sharded.py -q -m gpu -x > $OUT/r6_pytest_gpu_c.txt 2>&1; tail -12 $OUT/r6_pytest_gpu_c.txt
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/r6_bench_b.json 2> $OUT/r6_bench_b.err
python - <<PY
import json
d=json.loads(open("$OUT/r6_bench_b.json").read().strip().splitlines()[-1])
print("25/96: value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "ba", d["extra"]["ba_itrs2_us_p50"])
print(d["extra"]["bacore_update_us"])
PY
