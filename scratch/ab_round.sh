#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ba.py tests/test_gpu_sharded.py tests/test_gpu_entrypoints.py -x -q 2>&1 | tail -5 > gpurun_out/tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_tab.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp; O=/root/repo/gpurun_out; rm -rf $O/tab_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tab_trace -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; cp $(ls -t $(find $O/tab_trace -name "*kernel_stats.csv") | head -1) $O/tab_kernel_stats.csv
