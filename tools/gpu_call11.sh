#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 900 python tools/throughput_tune.py --top 24 --out gpurun_out/c11_tune_default.json 2> gpurun_out/c11_tt.err | tail -20
tail -3 gpurun_out/c11_tt.err | cut -c1-300
B="--no-cpu-baseline --no-latency --no-h2d --repeats 5"
for r in 1 2; do
  for leg in old new; do
    c=profiles/tune_cache.json; [ $leg = new ] && c=gpurun_out/c11_tune_default.json
    cp $c /tmp/ab_$leg.json
    timeout 400 python bench.py $B --tune-cache /tmp/ab_$leg.json > gpurun_out/c11_default_$leg$r.json 2> gpurun_out/c11_default_$leg$r.err
    python - gpurun_out/c11_default_$leg$r.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"])
except Exception as e:
    print(sys.argv[1], "NO RESULT", e)
PY
  done
done
