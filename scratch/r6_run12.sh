#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
for b in solve_wave_test solve_wave_x1 solve_wave_x2; do echo "== $b"; timeout 200 ./scratch/bin/$b 2>&1 | grep -A2 "P=63 n=378 w=4 extra=(-1,-1) spd=1\|P=63 n=378 w=8 \|P=63 n=378 w=4 extra=(-2\|P=63 n=378 w=6 \|P=24 n=144 w=4 extra=(-1,-1) spd=1" | grep "wave P\|wave:\|stages" | cut -c1-330; done > $OUT/r6_solver_experiment.txt 2>&1
cat $OUT/r6_solver_experiment.txt
