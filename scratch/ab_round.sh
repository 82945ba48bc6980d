#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
DBA_HIP_LIB=/root/repo/scratch/abl/libdba_hip_linprof.so timeout 120 python scratch/lin_prof.py > gpurun_out/lin_prof.log 2>&1
