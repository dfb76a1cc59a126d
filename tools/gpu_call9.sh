#!/bin/bash
# Round 5, ninth GPU call: host-fed pipeline with 2 x depth plans on `depth` streams; three / four batches in flight.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/c9_pipe.log 2>&1
echo "== pipeline tests: $(tail -1 gpurun_out/c9_pipe.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c9_pipe.log | head
for x in 1 2; do
  ICAF_PIPE_EXTRA_PLANS=$x timeout 400 python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/c9_h2d_$x.json 2> gpurun_out/c9_h2d_$x.err
  python - $x <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/c9_h2d_{sys.argv[1]}.json")); h = d.get("h2d_feed") or {}
    print("extra plan sets", sys.argv[1], "value", d["value"], "fwd_ms", d["forward_ms_per_batch"], "h2d", h.get("pairs_per_s_with_h2d"), h.get("min"), h.get("max"), h.get("pcie_gbs_achieved_in_loop"))
except Exception as e:
    print("h2d NO RESULT", e)
PY
done
for dpt in 2 3 4 2 3; do
  timeout 400 python bench.py --no-cpu-baseline --no-latency --no-h2d --repeats 5 --depth $dpt > gpurun_out/c9_depth$dpt.json 2> gpurun_out/c9_depth$dpt.err
  python - $dpt <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/c9_depth{sys.argv[1]}.json")); print("depth", sys.argv[1], "value", d["value"], d["value_min"], d["value_max"], "fwd-only", d["forward_only_pairs_per_s"])
except Exception as e:
    print("NO RESULT", e)
PY
done
for dpt in 2 3; do
  timeout 400 python bench.py --no-cpu-baseline --no-latency --no-h2d --repeats 3 --depth $dpt --model l --batch 32 --tune-cache profiles/tune_cache_c3_l_bf16_b32_640.json > gpurun_out/c9_c3_depth$dpt.json 2> gpurun_out/c9_c3_depth$dpt.err
  python - $dpt <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/c9_c3_depth{sys.argv[1]}.json")); print("c3 depth", sys.argv[1], "value", d["value"], d["value_min"], d["value_max"])
except Exception as e:
    print("NO RESULT", e)
PY
done
