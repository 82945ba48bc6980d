#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 ./scratch/solve_wave_test_bin > $OUT/r6_harness.txt 2>&1; echo "harness rc $?" >> $OUT/r6_harness.txt
grep -c MISMATCH $OUT/r6_harness.txt; grep -A2 "wave P" $OUT/r6_harness.txt | grep -B1 -A1 "us per solve" | grep -v "^--" | cut -c1-330 | head -120
