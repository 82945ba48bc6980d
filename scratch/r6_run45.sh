#!/bin/bash
export N_LIST=54
for rep in 1 2 3; do
DBA_BUILD_WG_TARGET=256 python scratch/build_n.py old 2>&1 | grep "48x64"
python scratch/build_n.py new 2>&1 | grep "48x64"
done
DBA_BUILD_DEBUG=1 python scratch/build_n.py new 2>&1 | grep "build:" | sort | uniq -c
DBA_BUILD_DEBUG=1 DBA_BUILD_WG_TARGET=256 python scratch/build_n.py old 2>&1 | grep "build:" | sort | uniq -c
