#!/bin/bash
# round-2 first GPU call: every GPU test, the 16-bit parity table, a bench line of the default workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_tests.sh
timeout 900 python tools/parity16.py --out gpurun_out/parity_16bit.json > gpurun_out/parity16.log 2>&1; tail -8 gpurun_out/parity16.log | cut -c1-900
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 3000 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
