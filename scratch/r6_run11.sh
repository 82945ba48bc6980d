#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export DBA_PARITY_REPORT=$OUT/r6_parity_b.jsonl; rm -f $DBA_PARITY_REPORT
timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_sharded.py -q -m gpu > $OUT/r6_pytest_gpu_d.txt 2>&1; tail -12 $OUT/r6_pytest_gpu_d.txt
for w in 25_96 64_512 32_122; do
timeout 300 python bench.py --window $w --steps 30 --warmup 8 --no-cpu-baseline --no-extras > $OUT/r6_x.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/r6_x.json").read().strip().splitlines()[-1])
print("$w: value", d["value"], "ms", d["ms_per_step"], "ba_itrs2", d["extra"].get("ba_itrs2_us_p50"))
PY
done
