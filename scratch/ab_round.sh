#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for w in 32_122 64_512; do timeout 300 python bench.py --no-cpu-baseline --window $w > gpurun_out/r01_bench_$w.json 2>/dev/null; done
