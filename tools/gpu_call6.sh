#!/bin/bash
# Round 5, sixth GPU call: fp32 token stream between DMFF iterations (tests; the 3-iteration configuration's 16-bit parity with it on / off), the host -> device
# copy in 1 / 2 / 4 slices on as many copy streams.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -s > gpurun_out/c6_dmff.log 2>&1
echo "== dmff fused tests: $(tail -1 gpurun_out/c6_dmff.log)"; grep -E "^(FAILED|ERROR)|fp32 stream" gpurun_out/c6_dmff.log | cut -c1-260 | head -24
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -x > gpurun_out/c6_model.log 2>&1
echo "== model / fullsize tests: $(tail -1 gpurun_out/c6_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c6_model.log | head
for r in 1 0; do
  ICAF_DMFF_RES32=$r timeout 600 python tools/parity16.py --only c4 --out gpurun_out/c6_parity_c4_res32_$r.json 2>/dev/null | cut -c1-900
done
for x in 1 2 4; do
  ICAF_PIPE_COPY_STREAMS=$x timeout 400 python bench.py --no-cpu-baseline --no-latency --repeats 3 > gpurun_out/c6_h2d_$x.json 2> gpurun_out/c6_h2d_$x.err
  python - $x <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/c6_h2d_{sys.argv[1]}.json")); h = d.get("h2d_feed") or {}
    print("copy streams", sys.argv[1], "value", d["value"], "fwd_ms", d["forward_ms_per_batch"], "h2d", h.get("pairs_per_s_with_h2d"), h.get("min"), h.get("max"), h.get("pcie_gbs_achieved_in_loop"))
except Exception as e:
    print("h2d NO RESULT", e)
PY
done
