#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for envs in "DBA_SCHUR_KERNEL=frame" "DBA_DETERMINISTIC=1" "DBA_WS_CACHE=0 DBA_BA_FUSE_UPDATE=0" "DBA_SCHUR_KERNEL=rows DBA_H_FULL=1" "DBA_SCHUR_WAVES=4 DBA_SCHUR_KERNEL=frame DBA_SCHUR_NCH=3"; do
  echo "=== $envs"
  env $envs timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_sharded.py tests/test_gpu_caller_sequence.py tests/test_replay.py -m gpu -q -x 2>&1 | tail -4
done
