#!/bin/bash
# round-6 call: new tests, bench line with the new fields, --force-gather A/B (one collective per group, own stream), fp32 bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -m gpu --timeout=600 -k "oracle_slice or rccl or pipeline_steps or several_batches" -p no:cacheprovider 2>&1 | tail -15
timeout 600 python bench.py --no-h2d > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6_bench_default.json"))
    print("default", d["value"], d["repeats"], d["timed_seconds"], d["roofline"]["bound"], d["roofline"]["frac_of_each_roof"], d.get("dmff_block"), d["cpu_baseline"].get("single_pair"))
except Exception as e:
    print("default FAILED", e); print(open("gpurun_out/r6_bench_default.err").read()[-3000:])
PY
for r in 1 2; do
  for g in "" "--force-gather"; do
    timeout 300 python bench.py --no-cpu-baseline --no-latency --no-h2d --min-timed-seconds 1.5 $g > gpurun_out/r6_fg_${r}_${g:+g}.json 2> gpurun_out/r6_fg_${r}_${g:+g}.err
    python -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r6_fg_${r}_${g:+g}.json')); print('gather' if '$g' else 'plain ', d['value'], d['value_min'], d['value_max'], d['config'].get('all_gather'))
except Exception as e: print('FAILED', e); print(open('gpurun_out/r6_fg_${r}_${g:+g}.err').read()[-2000:])
"
  done
done
timeout 900 python bench.py --dtype f32 --no-h2d --no-cpu-baseline --min-timed-seconds 1.5 > gpurun_out/r6_bench_fp32.json 2> gpurun_out/r6_bench_fp32.err
python -c "
import json
try:
    d=json.load(open('gpurun_out/r6_bench_fp32.json')); print('fp32', d['value'], d['forward_ms_per_batch'], d['roofline']['kernel'], d['roofline']['frac'])
except Exception as e: print('fp32 FAILED', e); print(open('gpurun_out/r6_bench_fp32.err').read()[-2000:])
"
