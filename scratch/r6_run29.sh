#!/bin/bash
OUT=$PWD/gpurun_out
timeout 900 python bench.py --no-cpu-baseline > $OUT/r6_bench_b.json 2> $OUT/r6_bench_b.err
python - <<PY
import json
d=json.load(open("$OUT/r6_bench_b.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("build_us_per_edge","build_frac_of_hbm_peak","build_into_slots_us_per_edge","build_into_slots_frac_of_hbm_peak","motion_filter_us","keyframe_cycle_us","zero_edit_churn_dba_update_per_s","zero_edit_shadow_build_us_per_edge"):
    print(k, d["extra"].get(k))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r6_trace_z
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r6_trace_z -- python /root/repo/scratch/build_ab.py prof > $OUT/r6_trace_z.log 2>&1
f=$(ls -t $(find $OUT/r6_trace_z -name "*kernel_stats.csv") | head -1)
grep "corr_build" $f | cut -c1-200
