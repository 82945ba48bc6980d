#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh r01
# Produces gpurun_out/<tag>_*: kernel-trace stats of the default bench command and the PMC passes
# (separate runs, --pmc never combined with sys/runtime traces) for the roofline kernel's HBM traffic.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc   # gpurun merges into an existing gpurun_out/: no stale traces
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
python $REPO/bench.py --steps 50 --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python $REPO/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_pmc -o pmc_$(echo $set | cut -c1-5) -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
cp $(ls -t $(find $OUT/${TAG}_trace -name "*kernel_stats.csv") | head -1) $OUT/${TAG}_kernel_stats.csv
python - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/${TAG}_pmc/*/*counter_collection.csv") + glob.glob("$OUT/${TAG}_pmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "corr_lookup_sheared" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
# bytes of a wide coalesced streaming read (16 B/lane -- the staging loads of this kernel), so it is doubled;
# WRITE_SIZE is calibrated on this kernel's known store volume (n*196*HW*2 bytes), see profiles/README.md
fetch = 2.0 * m.get("FETCH_SIZE", 0.0) * 1024.0
write = m.get("WRITE_SIZE", 0.0) * 1024.0
json.dump({"kernel": "corr_lookup_sheared_kernel<3>", "workload": "25_96", "FETCH_SIZE_KiB": m.get("FETCH_SIZE"),
           "WRITE_SIZE_KiB": m.get("WRITE_SIZE"), "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"),
           "hbm_read_bytes": fetch, "hbm_write_bytes": write, "traffic_bytes_per_launch": fetch + write,
           "launches_averaged": len(agg.get("FETCH_SIZE", []))}, open("$OUT/${TAG}_pmc_lookup.json", "w"), indent=1)
print(open("$OUT/${TAG}_pmc_lookup.json").read())
PY
cat $OUT/${TAG}_bench.json
