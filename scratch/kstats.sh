# kernel-trace stats of one bench window under an environment: bash scratch/kstats.sh <window> [VAR=value ...]
cd /tmp && export TMPDIR=/tmp
w=$1; shift
rm -rf /tmp/ks
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --window $w --steps 20 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(ls -t $(find /tmp/ks -name "*kernel_stats.csv") | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:12]:
    print("%-60s calls %4s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
