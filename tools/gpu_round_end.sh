#!/bin/bash
# Everything the round's committed numbers come from, in one GPU call: bench (with CPU baseline), rocprofv3 kernel stats
# of the same command, the two PMC passes (HBM traffic per kernel) and the per-layer profile.  Outputs under gpurun_out/;
# copy the summaries to profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/gpu_bench.sh "$@" 2>&1 | tail -30
cd $R && bash $R/tools/gpu_pmc.sh "$@" 2>&1 | tail -5
cd $R && timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -3 gpurun_out/layer_profile.txt
