#!/bin/bash
# Run ON THE GPU BOX: memory-pipeline stall counters of the lookup kernel (separate --pmc passes, kernel trace only).
# Every pass runs under `timeout`: the TA_* counter set made rocprofv3 abort and hang on this pool (it cost a
# 15-minute gpurun call), so it is left out.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_stalls
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o p$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "corr_lookup_sheared" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-48s %16.1f  (n=%d)" % (k, sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
