#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 300 ./scratch/mem_pattern3 > $OUT/r6_mem_pattern3.txt 2>&1; cat $OUT/r6_mem_pattern3.txt
