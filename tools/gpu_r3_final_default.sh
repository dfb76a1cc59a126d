# default workload with the final tune cache: PMC traffic / MFMA-busy passes, bench line + kernel stats, layer profiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
bash tools/gpu_pmc.sh 2>&1 | tail -1
cd $R && bash tools/gpu_pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1; grep -c mfma_util gpurun_out/pmc_sq.log
cd $R && bash tools/gpu_bench.sh 2>&1 | grep -v "^\"\|^W2026" | tail -2 | cut -c1-200
cd $R && timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -1 gpurun_out/layer_profile.txt
cd $R && ICAF_DMFF_FUSE=0 timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile_plain.txt 2>/dev/null; head -1 gpurun_out/layer_profile_plain.txt
python -c "
import json
d = json.load(open('gpurun_out/bench.json')); r = d['roofline']; print(d['value'], d['value_min'], d['value_max'], d['forward_ms_per_batch'], d['forward_only_pairs_per_s'], d['forward_only_pairs_per_s_one_in_flight'], r['kernel'], r['frac'], r['traffic'], r['algorithmic_bytes_per_launch'])
"
