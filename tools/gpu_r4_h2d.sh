#!/bin/bash
# the uint8 host feed: copy stream at high priority; default hardware queues vs 8 vs 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
run () { timeout 300 python bench.py --no-cpu-baseline --no-latency --repeats 3 > gpurun_out/h2d_$1.json 2> gpurun_out/h2d_$1.err
  python -c "
import json; d=json.load(open('gpurun_out/h2d_$1.json')); print('$1 value', d['value'], 'fwd_ms', d['forward_ms_per_batch'], 'h2d', d['h2d_feed']['pairs_per_s_with_h2d'], d['h2d_feed']['pcie_gbs_achieved_in_loop'], d['h2d_feed']['pcie_gbs_copy_alone'])"; }
run default
GPU_MAX_HW_QUEUES=8 run q8
GPU_MAX_HW_QUEUES=2 run q2
run default2
