# kernel-trace of the motion filter's unit (scratch/motion_prof.py): which launches one frame consists of
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mk -- python $GRAFT_REPO_ROOT/scratch/motion_prof.py 2>&1 | grep "motion filter"
f=$(ls -t $(find /tmp/mk -name "*kernel_stats.csv") | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:8]:
    print("%-70s calls %4s avg %8.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
t=$(ls -t $(find /tmp/mk -name "*kernel_trace.csv") | head -1)
python - <<PY
import csv
rows = sorted(csv.DictReader(open("$t")), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-12:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%-60s start %8.2f us  dur %7.2f us" % (r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
