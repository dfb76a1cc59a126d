#!/bin/bash
# One short GPU call (about 40 s): tests of the kernels touched last (Detect decode, nearest up-sampling, DMFF pooling, pipeline), a per-launch
# profile with the rows of those kernels printed, and the default bench twice (the spread of one box).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
t0=$(date +%s)
timeout 240 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu -k "detect_decode or sppf_upsample or dmff_pool_tokens or nearest_pre_term or pipeline_steps or batches_in_flight" --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/ab_tests.log 2>&1
echo "== tests: $(tail -1 gpurun_out/ab_tests.log) [$(( $(date +%s) - t0 )) s]"; grep -E "^(FAILED|ERROR)" gpurun_out/ab_tests.log | head
timeout 200 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "golden or graph_replay or other_shape or low_precision or map50" --timeout=180 --tb=short -p no:cacheprovider > gpurun_out/ab_model.log 2>&1
echo "== model: $(tail -1 gpurun_out/ab_model.log) [$(( $(date +%s) - t0 )) s]"; grep -E "^(FAILED|ERROR)" gpurun_out/ab_model.log | head
timeout 120 python tools/layer_profile.py --autotune > gpurun_out/ab_layers.txt 2>&1
grep -E "total|detect_decode|upsample_nearest|sppf_pool|dmff_pool_tokens" gpurun_out/ab_layers.txt
echo "[$(( $(date +%s) - t0 )) s]"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], {k: d.get(k) for k in ("value", "value_min", "value_max", "forward_only_pairs_per_s", "forward_ms_per_batch")})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
timeout 150 python bench.py --no-cpu-baseline --repeats 5 > gpurun_out/ab_default.json 2> gpurun_out/ab_default.err; show gpurun_out/ab_default.json
timeout 150 python bench.py --no-cpu-baseline --repeats 5 > gpurun_out/ab_default2.json 2> gpurun_out/ab_default2.err; show gpurun_out/ab_default2.json
echo "[$(( $(date +%s) - t0 )) s]"
