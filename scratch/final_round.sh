#!/bin/bash
# full GPU suite + the round's profile (run through gpurun from the repo root)
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/tests_full.log
bash tools/profile_round.sh r01 > gpurun_out/profile_round.log 2>&1
