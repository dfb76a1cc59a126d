R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json --loops 3 --height 512 --width 640 --batch 64 > gpurun_out/bench_c4_s_bf16_b64_512x640_loops3.json 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_c3_l_bf16_b32_640.json --model l --batch 32 > gpurun_out/bench_c3_l_bf16_b32_640.json 2>/dev/null
python -c "
import json
for n in ('c4_s_bf16_b64_512x640_loops3', 'c3_l_bf16_b32_640'):
    d = json.load(open(f'gpurun_out/bench_{n}.json')); r = d['roofline']; print(n, d['value'], d['value_min'], d['value_max'], d['forward_ms_per_batch'], r['kernel'], r['frac'], r['traffic'])
"
