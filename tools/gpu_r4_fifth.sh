#!/bin/bash
# Round 4, fifth GPU call: whole-graph refinement of the default workload's tile choices on this round's kernels, then a same-box A/B of the two caches.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python tools/graph_tune.py --top 40 --eps 0.002 --out gpurun_out/gt_default.json 2>&1 | grep -E "start|final|->" | tail -n 20
for c in profiles/tune_cache.json gpurun_out/gt_default.json profiles/tune_cache.json gpurun_out/gt_default.json; do
  ICAF_AB_TUNE=$R/$c timeout 300 python tools/probes/ab_lib.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$c', [round(x,4) for x in d['forward_ms']])"
done
timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -1 gpurun_out/layer_profile.txt
