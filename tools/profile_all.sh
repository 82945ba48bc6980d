#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_all.sh r02
# One pass over everything profiles/ holds for a round:
#   * bench line + kernel-trace stats + HBM-traffic PMC passes of the roofline kernel for the three bench windows
#     (tools/profile_round.sh);
#   * SQ instruction / MFMA counters for the lookup, build, linearisation and solver kernels (separate --pmc passes, kernel
#     trace only: --pmc is never combined with sys / runtime trace domains);
#   * kernel-trace stats of the on-the-fly correlation (scratch/altcorr_bench.py);
#   * the HBM ceilings of the box (scratch/hbm_ceiling.py, scratch/bin/mem_pattern) next to them.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
bash tools/profile_round.sh $TAG 25_96 corr_lookup_rowtile > $OUT/${TAG}_round_25_96.log 2>&1
bash tools/profile_round.sh $TAG 64_512 corr_lookup_rowtile > $OUT/${TAG}_round_64_512.log 2>&1
bash tools/profile_round.sh $TAG 32_122 corr_lookup_resident > $OUT/${TAG}_round_32_122.log 2>&1
bash tools/profile_round.sh $TAG 9_36_55x55 corr_lookup_resident > $OUT/${TAG}_round_9_36_55x55.log 2>&1
bash tools/profile_round.sh $TAG 10_54_48x64 corr_lookup_rowtile > $OUT/${TAG}_round_10_54_48x64.log 2>&1
cd /tmp && export TMPDIR=/tmp
SQ=$OUT/${TAG}_sq
rm -rf $SQ; mkdir -p $SQ
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $SQ -o p$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $SQ/log$i.txt 2>&1
done
# the per-source-frame Schur kernel only runs on dense windows: its instruction / matrix-core counters from the 64-KF window
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $SQ -o p$i -- python $REPO/bench.py --window 64_512 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $SQ/log$i.txt 2>&1
done
# HBM write traffic of the build kernel (its own pass)
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $SQ -o pw -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $SQ/logw.txt 2>&1
python - > $OUT/${TAG}_sq_counters.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
names = ("corr_lookup_rowtile", "corr_lookup_resident", "corr_build_fused", "ba_linearize", "ba_schur_gram", "ba_schur_kernel", "ba_solve_wave", "ba_solve_tile", "ba_solve_band", "ba_update", "ba_prepare")
for f in glob.glob("$SQ/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for nm in names:
            if nm in k: agg[(nm, int(r.get("Grid_Size") or 0), r["Counter_Name"])].append(float(r["Counter_Value"]))
print("# per launch, averaged over the launches of one grid size (threads): bench.py --steps 3 --warmup 1 on the 25 KF / 96 edges window and,")
print("# for the matrix-core Schur passes, --window 64_512 --steps 2 (the larger grids); rocprofv3 --pmc, kernel trace only")
for k in sorted(agg): print("%-22s grid %-9d %-32s %16.1f  (n=%d)" % (k[0], k[1], k[2], sum(agg[k]) / len(agg[k]), len(agg[k])))
PY
cat $OUT/${TAG}_sq_counters.txt | head -80
# on-the-fly correlation
rm -rf $OUT/${TAG}_alt_trace
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_alt_trace -- python $REPO/scratch/altcorr_bench.py > $OUT/${TAG}_altcorr_bench.txt 2>&1
cp $(ls -t $(find $OUT/${TAG}_alt_trace -name "*kernel_stats.csv") | head -1) $OUT/${TAG}_altcorr_kernel_stats.csv
# ceilings
python $REPO/scratch/hbm_ceiling.py > $OUT/${TAG}_hbm_ceiling.txt 2>&1
[ -x $REPO/scratch/bin/mem_pattern2 ] && $REPO/scratch/bin/mem_pattern2 96 > $OUT/${TAG}_mem_pattern2_96.txt 2>&1 && $REPO/scratch/bin/mem_pattern2 384 > $OUT/${TAG}_mem_pattern2_384.txt 2>&1
# per-CU rates of the vector memory pipe and of LDS, dependent-chain latencies of the solver's f64 ops, solver stage times
[ -x $REPO/scratch/bin/cu_rates ] && $REPO/scratch/bin/cu_rates > $OUT/${TAG}_cu_rates.txt 2>&1
[ -x $REPO/scratch/bin/lds_rates ] && $REPO/scratch/bin/lds_rates > $OUT/${TAG}_lds_rates.txt 2>&1
[ -x $REPO/scratch/bin/f64_latency ] && $REPO/scratch/bin/f64_latency > $OUT/${TAG}_f64_latency.txt 2>&1
# (window solver: per-wave stage times + the matrix instruction's latency; then the register-tile / skyline kernels' stages)
{ [ -x $REPO/scratch/bin/solve_wave_test ] && timeout 120 $REPO/scratch/bin/solve_wave_test 2>&1 | grep -A3 "P=24 n=144 w=4 extra=(-1,-1) spd=1\|P=29 n=174\|P=63 n=378 w=4 extra=(-1,-1) spd=1\|P=40 n=240\|n=144 w=5" | grep -v "^--\|P=64 n=384"
  [ -x $REPO/scratch/bin/solve_wave_two_fronts ] && echo "--- shelved two-front variant (scratch/ba_solve_wave_two_fronts.hip)" && timeout 120 $REPO/scratch/bin/solve_wave_two_fronts 2>&1 | grep -A4 "P=24 n=144 w=4 extra=(-1,-1) spd=1\|P=40 n=240" | grep -v "^--\|P=64 n=384"
  [ -x $REPO/scratch/bin/mfma_f64_lat ] && echo "--- scratch/mfma_f64_lat.hip" && timeout 60 $REPO/scratch/bin/mfma_f64_lat 2>&1 | grep "cycles per link"
  [ -x $REPO/scratch/bin/solve_prof ] && echo "--- register-tile / skyline kernels (scratch/solve_tile_test.hip)" && (HARNESS_BAND=1 timeout 120 $REPO/scratch/bin/solve_prof; timeout 120 $REPO/scratch/bin/solve_prof) 2>&1 | grep -A3 "n=144 band= 24 spd=1\|n=186\|n=378 band= 36"
} > $OUT/${TAG}_solver_stages.txt
timeout 120 python $REPO/scratch/build_ab.py $TAG > $OUT/${TAG}_build_shapes.txt 2>&1
ls -la $OUT | grep ${TAG}_ | head -50
