#!/bin/bash
# deep-prefetch register pipeline (igemm ids 51 / 52 / 54): bit-identity at full grid, then a fresh tuning run against the committed cache
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "every_launch or bench_configuration" --timeout=600 --tb=short -p no:cacheprovider > gpurun_out/regd_tests.log 2>&1
echo "== bit identity: $(tail -1 gpurun_out/regd_tests.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/regd_tests.log | head
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_cache.json 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline --tune-cache gpurun_out/tune_regd.json > gpurun_out/bench_regd.json 2> gpurun_out/bench_regd.err
python - <<'PY'
import json
for n in ("bench_cache", "bench_regd"):
    d = json.load(open(f"gpurun_out/{n}.json"))
    print(n, d["value"], d["forward_only_pairs_per_s"], d["forward_ms_per_batch"], d["forward_roofline"]["mfma_frac"])
    print("   ", {k: v["ms_per_step"] for k, v in list(d["kernels"].items())[:8]})
t = json.load(open("gpurun_out/tune_regd.json"))
import collections
print(collections.Counter(v for _, v in t))
PY
timeout 300 python tools/layer_profile.py --sweep 2>/dev/null | tail -60 > gpurun_out/sweep.txt; tail -45 gpurun_out/sweep.txt | cut -c1-200
