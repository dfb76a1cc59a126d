#!/bin/bash
# the four bench lines (default workload with CPU baseline; per-GPU shards of BASELINE configs 3 / 4 / 5) with the committed PMC summaries
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
timeout 600 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_c3_l_bf16_b32_640.json --model l --batch 32 > gpurun_out/bench_c3_l_bf16_b32_640.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json --loops 3 --height 512 --width 640 --batch 64 > gpurun_out/bench_c4_s_bf16_b64_512x640_loops3.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_c5_l_vedai_f16_b16_1280.json --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3 > gpurun_out/bench_c5_l_vedai_f16_b16_1280.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench", "bench_c3_l_bf16_b32_640", "bench_c4_s_bf16_b64_512x640_loops3", "bench_c5_l_vedai_f16_b16_1280"):
    d = json.load(open(f"gpurun_out/{n}.json"))
    print(n, {k: d[k] for k in ("value", "value_min", "value_max", "ms_per_step", "forward_only_pairs_per_s", "forward_ms_per_batch", "nms_ms_per_batch_standalone")}, d["forward_roofline"]["mfma_frac"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
