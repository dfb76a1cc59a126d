#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "nms" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/q_nms.log 2>&1
echo "== nms: $(tail -1 gpurun_out/q_nms.log)"; grep -E "^(FAILED|ERROR)|Error|Mismatch" gpurun_out/q_nms.log | head
timeout 600 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_q.json"))
print({k: d[k] for k in ("value", "value_min", "value_max", "ms_per_step", "forward_only_pairs_per_s", "forward_ms_per_batch", "nms_ms_per_batch_standalone", "forward_roofline")})
PY
tail -3 gpurun_out/bench_q.err
