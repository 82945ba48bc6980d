#!/bin/bash
timeout 200 ./scratch/bin/solve_wave_test 2>&1 | grep -c "MISMATCH\|launch error"
timeout 200 ./scratch/bin/solve_wave_test 2>&1 | grep -A1 "P=63 n=378 w=4 extra\|P=63 n=378 w=6\|P=49" | grep "wave:"
