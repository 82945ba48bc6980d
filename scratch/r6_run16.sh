#!/bin/bash
for i in 1 2 3; do
for lib in scratch/libdba_hip_base.so dba-fusion_amd/lib/libdba_hip.so; do DBA_HIP_LIB=$PWD/$lib python scratch/bacore_ab.py 25_96 300 2>&1 | tail -1; done
done
