#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
O=$REPO/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in weak strong; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --scaling $mode --no-cpu-baseline 2>$O/r03_two_ranks_$mode.err | grep "^{" > $O/r03_bench_2ranks_one_gpu_gloo_$mode.json
  python -c "
import json; d=json.load(open('$O/r03_bench_2ranks_one_gpu_gloo_$mode.json')); print('$mode', d['value'], d['unit'], d['ms_per_step'], d['config']['edges'], d['config']['exchange'], d['extra'].get('edge_throughput_vs_1gpu'), d['extra'].get('edges_per_s'))"
done
timeout 300 python bench.py --scaling weak --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/r03_bench_weak_n1.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r03_bench_weak_n1.json')); print('weak n1', d['value'], d['ms_per_step'], d['extra']['edges_per_s'], d['extra']['ba_itrs2_us_p50'], d['roofline']['avg_launch_ms'])"
