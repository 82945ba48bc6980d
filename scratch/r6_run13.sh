#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x -k captured --durations=5 2>&1 | grep -v "^E    *[0-9-]\|^E   *\[" | tail -50
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu -x 2>&1 | tail -3
