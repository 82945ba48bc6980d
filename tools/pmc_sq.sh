#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o p$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for nm in ("corr_lookup_sheared", "ba_linearize", "ba_solve_tile"):
            if nm in k: agg[(nm, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print(k, sum(agg[k]) / len(agg[k]), len(agg[k]))
PY
