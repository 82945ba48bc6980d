#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "bacore or fusion or caller_sequence or compiled_adapter or sharded" 2>&1 | tail -4
timeout 900 python bench.py --no-cpu-baseline > $OUT/r6_bench_a.json 2> $OUT/r6_bench_a.err
python - <<PY
import json
d=json.load(open("$OUT/r6_bench_a.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("bacore_update_us","bacore_update_us_10kf_54edges_48x64_sensor_depth","bacore_gtsam_handover_us","ba_itrs2_us_p50","motion_filter_us","build_us_per_edge"):
    print(k, d["extra"].get(k))
PY
