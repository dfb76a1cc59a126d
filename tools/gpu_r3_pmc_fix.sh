# default workload + config 4: PMC traffic / MFMA-busy passes and the bench lines again (the first summaries merged the two streaming-GEMM tiles under one name)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
bash tools/gpu_pmc.sh 2>&1 | tail -1
cd $R && bash tools/gpu_pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1; grep -c mfma_util gpurun_out/pmc_sq.log
cd $R && bash tools/gpu_bench.sh 2>&1 | grep -v "^\"\|^W2026" | tail -2 | cut -c1-300
name=c4_s_bf16_b64_512x640_loops3
cd $R && PMC_NAME=$name bash tools/gpu_pmc.sh --tune-cache $R/profiles/tune_cache_$name.json --loops 3 --height 512 --width 640 --batch 64 2>&1 | tail -1
cd $R && timeout 900 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_$name.json --loops 3 --height 512 --width 640 --batch 64 > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
python -c "
import json
for n in ('bench', 'bench_$name'):
    d = json.load(open(f'gpurun_out/{n}.json')); r = d['roofline']; print(n, d['value'], d['forward_ms_per_batch'], r['kernel'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'])
"
