#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh r02 [window] [kernel-substring]
# Produces gpurun_out/<tag>_*: the bench line, kernel-trace stats of the same command, and the PMC passes (separate
# runs, --pmc never combined with sys/runtime traces) for the roofline kernel's HBM traffic and instruction mix.
set -u
TAG=${1:-r03}
WIN=${2:-25_96}
KERN=${3:-corr_lookup_rowtile}
SFX=""; [ "$WIN" != "25_96" ] && SFX="_$WIN"
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_trace$SFX $OUT/${TAG}_pmc$SFX   # gpurun merges into an existing gpurun_out/: no stale traces
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--window $WIN --steps 60 --warmup 12"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace$SFX -- python $REPO/bench.py $ARGS --no-cpu-baseline --no-extras > $OUT/${TAG}_trace$SFX.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${TAG}_pmc$SFX -o pmc_$(echo $set | cut -c1-5) -- python $REPO/bench.py --window $WIN --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
done
cp $(ls -t $(find $OUT/${TAG}_trace$SFX -name "*kernel_stats.csv") | head -1) $OUT/${TAG}_kernel_stats$SFX.csv
python - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$REPO")
import bench
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/${TAG}_pmc$SFX/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
# MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
# bytes of a wide coalesced streaming read (16 B/lane -- the staging loads of this kernel), so it is doubled;
# WRITE_SIZE is calibrated on this kernel's known store volume (n*196*HW*2 bytes), see profiles/README.md
fetch = 2.0 * m.get("FETCH_SIZE", 0.0) * 1024.0
write = m.get("WRITE_SIZE", 0.0) * 1024.0
json.dump({"kernel": "$KERN", "workload": "$WIN", "copies": 3, "kernel_source_sha": bench.source_hash(bench.LOOKUP_SOURCES),
           "FETCH_SIZE_KiB": m.get("FETCH_SIZE"), "WRITE_SIZE_KiB": m.get("WRITE_SIZE"), "TCC_HIT_sum": m.get("TCC_HIT_sum"),
           "TCC_MISS_sum": m.get("TCC_MISS_sum"), "hbm_read_bytes": fetch, "hbm_write_bytes": write,
           "traffic_bytes_per_launch": fetch + write, "launches_averaged": len(agg.get("FETCH_SIZE", []))},
          open("$OUT/${TAG}_pmc_lookup$SFX.json", "w"), indent=1)
print(open("$OUT/${TAG}_pmc_lookup$SFX.json").read())
PY
# the bench line last: it reads the traffic per launch from the round's PMC file under profiles/ (copied there on this box;
# the builder commits the same file from gpurun_out/)
cp $OUT/${TAG}_pmc_lookup$SFX.json $REPO/profiles/${TAG}_pmc_lookup$SFX.json
timeout 600 python $REPO/bench.py $ARGS > $OUT/${TAG}_bench$SFX.json 2> $OUT/${TAG}_bench$SFX.err
cat $OUT/${TAG}_bench$SFX.json
