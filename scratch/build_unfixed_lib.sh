#!/bin/bash
# experiment only: the library with round 5's hand-over race compiled back in (the new taker of flagE[NT-1] does not wait for the
# old holder's last announcement), to see whether tests/native/solve_cold + tests/test_gpu_solve_cold.py catch it.
#   bash scratch/build_unfixed_lib.sh   ->  scratch/libdba_hip_unfixed.so   (never loaded by the product or the suite)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/unfixed
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -DWV_TEST_UNFIXED_HANDOVER \
    -c dba-fusion_amd/csrc/ba_solve_wave.hip -o build/unfixed/ba_solve_wave.o
objs=$(ls build/gfx950/*.o | grep -v ba_solve_wave.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/libdba_hip_unfixed.so $objs build/unfixed/ba_solve_wave.o
ls -la scratch/libdba_hip_unfixed.so
