#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
for ew in 2 1; do for w in 64_512 25_96 32_122; do
DBA_LIN_EW=$ew timeout 300 python bench.py --window $w --steps 30 --warmup 8 --no-cpu-baseline --no-extras > $OUT/r6_ew.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("$OUT/r6_ew.json").read().strip().splitlines()[-1])
print("EW=$ew $w: value", d["value"], "ms", d["ms_per_step"], "ba_itrs2", d["extra"].get("ba_itrs2_us_p50"))
PY
done; done 2>&1 | tee $OUT/r6_lin_ew.txt
