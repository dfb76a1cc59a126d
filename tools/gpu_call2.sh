#!/bin/bash
# round-2 second GPU call: the rewritten NMS and the fused DMFF block kernels — tests, bench line, per-layer profile, rocprof kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "nms or match_predictions" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/c2_nms.log 2>&1
echo "== nms: $(tail -1 gpurun_out/c2_nms.log)"; grep -E "^(FAILED|ERROR)|Error|error:|Mismatch|mismatch" gpurun_out/c2_nms.log | head -20
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -s > gpurun_out/c2_dmff.log 2>&1
echo "== dmff fused: $(tail -1 gpurun_out/c2_dmff.log)"; grep -E "^(FAILED|ERROR)|Error" gpurun_out/c2_dmff.log | head -20; grep "fused" gpurun_out/c2_dmff.log | head -40
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_frontends.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider > gpurun_out/c2_model.log 2>&1
echo "== model/frontends/pipeline/fullsize: $(tail -1 gpurun_out/c2_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c2_model.log | head -20; grep -E "AssertionError|Error:|assert " gpurun_out/c2_model.log | sort | uniq -c | sort -rn | head -12
timeout 600 python -m pytest tests/test_gpu_parity16.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s -k "c2 or c4 or map50" > gpurun_out/c2_parity.log 2>&1
echo "== parity16 (c2, c4, map): $(tail -1 gpurun_out/c2_parity.log)"; grep -E "^(FAILED|ERROR)|AssertionError" gpurun_out/c2_parity.log | head; grep '"dtype"' gpurun_out/c2_parity.log | cut -c1-700
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print({k: d[k] for k in ("value", "value_min", "value_max", "ms_per_step", "forward_only_pairs_per_s", "forward_ms_per_batch", "forward_roofline")})
for k, v in list(d["kernels"].items())[:14]: print(k, v)
PY
tail -3 gpurun_out/bench.err
timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -1 gpurun_out/layer_profile.txt; grep -n "dmff\|qkv\|attn\|nms" gpurun_out/layer_profile.txt | head -20
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -30
