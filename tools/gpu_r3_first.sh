#!/bin/bash
# Round 3, first GPU call: the whole GPU suite (new: rect golden, planted-detector mAP, measured fp32 errors, detection block), a layer
# profile and two quick bench lines (one / two batches in flight) on the same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/parity_fp32.jsonl gpurun_out/parity_16bit_tests.jsonl
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s > gpurun_out/r3_tests.log 2>&1
echo "== gpu tests: $(tail -1 gpurun_out/r3_tests.log)"
grep -E "^(FAILED|ERROR)" gpurun_out/r3_tests.log | head -30
grep -E "AssertionError|Error:|assert " gpurun_out/r3_tests.log | sort | uniq -c | sort -rn | head -20
timeout 300 python tools/layer_profile.py --autotune > gpurun_out/r3_layer_profile.txt 2> gpurun_out/r3_layer_profile.err; head -1 gpurun_out/r3_layer_profile.txt
for d in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --repeats 5 --depth $d > gpurun_out/r3_bench_d$d.json 2> gpurun_out/r3_bench_d$d.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r3_bench_d$d.json"))
    print("depth $d:", {k: d.get(k) for k in ("value", "ms_per_step", "forward_ms_per_batch", "forward_only_pairs_per_s_one_in_flight", "latency_ms_b1", "nms_ms_per_batch_standalone")}, d.get("latency_b1"))
except Exception as e:
    print("bench depth $d failed", e)
PY
  tail -2 gpurun_out/r3_bench_d$d.err
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
