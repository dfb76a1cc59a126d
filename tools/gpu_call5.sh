#!/bin/bash
# Round 5, fifth GPU call: attention K / V^T staging with four keys per item (tests + times), host-fed pipeline with 1 / 2 / 3 extra plans.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cross_attention" --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/c5_attn.log 2>&1
echo "== attention tests: $(tail -1 gpurun_out/c5_attn.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c5_attn.log | head
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py tests/test_gpu_model.py tests/test_gpu_pipeline.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/c5_model.log 2>&1
echo "== dmff / model / pipeline tests: $(tail -1 gpurun_out/c5_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c5_model.log | head
for m in s l; do echo "model $m"; timeout 200 python tools/probes/dmff_levels.py $m 2>/dev/null | grep three | sed 's/.*| //'; done
for x in 1 2 3; do
  ICAF_PIPE_EXTRA_PLANS=$x timeout 400 python bench.py --no-cpu-baseline --no-latency --repeats 3 > gpurun_out/c5_h2d_$x.json 2> gpurun_out/c5_h2d_$x.err
  python - $x <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/c5_h2d_{sys.argv[1]}.json")); h = d.get("h2d_feed") or {}
    print("extra plans", sys.argv[1], "value", d["value"], "fwd_ms", d["forward_ms_per_batch"], "h2d", h.get("pairs_per_s_with_h2d"), h.get("min"), h.get("max"), h.get("pcie_gbs_achieved_in_loop"))
except Exception as e:
    print("h2d NO RESULT", e)
PY
done
