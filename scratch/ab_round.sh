#!/bin/bash
cd /tmp; export TMPDIR=/tmp; O=/root/repo/gpurun_out; rm -rf $O/sr_trace
DBA_HIP_LIB=/root/repo/scratch/abl/libdba_hip_schurred.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/sr_trace -- python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; cp $(ls -t $(find $O/sr_trace -name "*kernel_stats.csv") | head -1) $O/sr_kernel_stats.csv
