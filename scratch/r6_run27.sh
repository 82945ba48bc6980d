#!/bin/bash
python scratch/build_ab.py w16 2>&1 | grep -v amdgpu.ids
DBA_BUILD_WAVES=8 python scratch/build_ab.py w8 2>&1 | grep -v amdgpu.ids
python scratch/build_ab.py w16 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_shapes.py tests/test_gpu_corr_slots.py -q -m gpu -x 2>&1 | tail -4
