#!/bin/bash
# Round 5, eighth GPU call: tools/graph_tune.py (whole-graph replay time as the clock) from the committed choices, then the bench with the refined cache against
# the committed one (the refinement optimises ONE forward's latency; the bench keeps two in flight — adopt only what the bench confirms).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
B="--no-cpu-baseline --no-latency --no-h2d --repeats 5"
timeout 600 python tools/graph_tune.py --top 30 --out gpurun_out/c8_tune_default.json 2>/dev/null | grep -v "^W2026" | tail -12
timeout 600 python tools/graph_tune.py --model l --top 30 --seed-cache profiles/tune_cache_c3_l_bf16_b32_640.json --out gpurun_out/c8_tune_c3.json 2>/dev/null | tail -12
timeout 600 python tools/graph_tune.py --loops 3 --height 512 --width 640 --batch 64 --top 30 --seed-cache profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json --out gpurun_out/c8_tune_c4.json 2>/dev/null | tail -8
ab () {  # tag, committed cache, new cache, args
  tag=$1; old=$2; new=$3; shift 3
  for r in 1 2; do
    for leg in old new; do
      c=$old; [ $leg = new ] && c=$new
      cp $c /tmp/ab_$leg.json
      timeout 400 python bench.py $B --tune-cache /tmp/ab_$leg.json "$@" > gpurun_out/c8_${tag}_$leg$r.json 2> gpurun_out/c8_${tag}_$leg$r.err
      python - gpurun_out/c8_${tag}_$leg$r.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"])
except Exception as e:
    print(sys.argv[1], "NO RESULT", e)
PY
    done
  done
}
ab default profiles/tune_cache.json gpurun_out/c8_tune_default.json
ab c3 profiles/tune_cache_c3_l_bf16_b32_640.json gpurun_out/c8_tune_c3.json --model l --batch 32
ab c4 profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json gpurun_out/c8_tune_c4.json --loops 3 --height 512 --width 640 --batch 64
