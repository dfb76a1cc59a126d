#!/bin/bash
# Round 5, fourth GPU call: fp32 instantiation of the three-launch DMFF kernels vs oracle / reference goldens; attention split heuristic; host-fed pipeline
# with one forward stream per plan.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 600 python -m pytest tests/test_gpu_dmff_fused.py tests/test_gpu_pipeline.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -s > gpurun_out/c4_dmff.log 2>&1
echo "== dmff fused + pipeline tests: $(tail -1 gpurun_out/c4_dmff.log)"; grep -E "^(FAILED|ERROR)|fp32 wide|three-launch" gpurun_out/c4_dmff.log | head -20
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "cross_attention" --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/c4_attn.log 2>&1
echo "== attention tests: $(tail -1 gpurun_out/c4_attn.log)"
for m in s l; do echo "model $m"; timeout 200 python tools/probes/dmff_levels.py $m 2>/dev/null | grep three | sed 's/.*| //'; done
ICAF_PERS_GEMM=0 timeout 400 python bench.py --no-cpu-baseline --no-latency --repeats 5 > gpurun_out/c4_h2d.json 2> gpurun_out/c4_h2d.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/c4_h2d.json")); print("h2d", d["value"], d["forward_ms_per_batch"], json.dumps(d.get("h2d_feed"))[:400])
except Exception as e:
    print("h2d NO RESULT", e)
PY
