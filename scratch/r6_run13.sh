#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for t in 512 256 128; do echo "== DBA_BUILD_WG_TARGET=$t"; DBA_BUILD_WG_TARGET=$t python scratch/motion_prof.py 2>&1 | tail -1; DBA_BUILD_WG_TARGET=$t python scratch/build_ab.py r6 2>&1 | tail -6; done > $OUT/r6_build_wg.txt 2>&1
cat $OUT/r6_build_wg.txt
timeout 600 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_shapes.py tests/test_gpu_corr_slots.py -q -m gpu 2>&1 | tail -2
