#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
{
echo "== twist"; timeout 120 scratch/bin/solve_twist 2>&1 | grep -v "block:\|band :\|ticks" | grep -v "^band"
} > gpurun_out/solve_ab.log 2>&1
