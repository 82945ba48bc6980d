#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out/b13; rm -rf $O; mkdir -p $O
echo "=== f32 frame forced"; DBA_SCHUR_MFMA=f32 DBA_SCHUR_KERNEL=frame timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -4
run() {
  local label=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --window $w --steps 60 --warmup 12 --no-cpu-baseline --no-extras 2>$O/err_${label}_$w.txt | python -c "
import json,sys
try:
    d=json.load(sys.stdin); print('$label $w', d['value'], 'upd/s  ms', d['ms_per_step'], 'lookup', d['roofline']['avg_launch_ms'], 'ba', d['extra']['ba_itrs2_us_p50'])
except Exception as e: print('$label $w FAILED', e)"
}
run f64 64_512 X=1
run f32 64_512 DBA_SCHUR_MFMA=f32
run f64 64_512 X=1
run f32 64_512 DBA_SCHUR_MFMA=f32
run f32w4 64_512 DBA_SCHUR_MFMA=f32 DBA_SCHUR_WAVES=4
run f32nch8 64_512 DBA_SCHUR_MFMA=f32 DBA_SCHUR_NCH=8
python - <<PY
import json
for l in open("$REPO/gpurun_out/parity_report.jsonl"):
    d=json.loads(l)
    if "64kf" in d.get("test","") and "dt_m" in d and "matches_oracle" in d["test"]:
        last=d
print("64kf parity (last, f32 forced run):", last["test"].split("::")[-1], last["dt_m"], last["dr_rad"], last["depth_max_err_over_scale"])
PY
