#!/bin/bash
# pre-term GEMMs on the 8-wavefront tile: bit identity of every configuration, then same-box A/B of the bench (committed cache / fresh
# tuning / fresh tuning with the folded up-sampling)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "every_launch or bench_configuration" --timeout=600 --tb=short -p no:cacheprovider > gpurun_out/pre_tests.log 2>&1
echo "== bit identity: $(tail -1 gpurun_out/pre_tests.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/pre_tests.log | head
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --tune-cache gpurun_out/tune_pre.json > gpurun_out/bench_b.json 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --fold-upsample --tune-cache gpurun_out/tune_pre_fold.json > gpurun_out/bench_c.json 2>/dev/null
python - <<'PY'
import json, collections
for n in ("bench_a", "bench_b", "bench_c"):
    d = json.load(open(f"gpurun_out/{n}.json"))
    print(n, d["value"], d["forward_only_pairs_per_s"], d["forward_ms_per_batch"])
t = json.load(open("gpurun_out/tune_pre.json"))
print("pre launches:", [(k, v) for k, v in t if k[16]])
t = json.load(open("gpurun_out/tune_pre_fold.json"))
print("pre launches (fold):", [(k[0], k[1], k[2], v) for k, v in t if k[16]])
PY
