#!/bin/bash
# Round-4 evidence, refresh of the two yolov5s workloads (default, config 4) after a change that only they see (their tune caches; stem2 / icaf_bottleneck):
# same steps as tools/gpu_round4_evidence.sh for those two (configs 3 / 5 keep their files: nothing on their paths changed since that call).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
cd $R && bash tools/gpu_pmc.sh 2>&1 | tail -2
cd $R && SQ_INSTS=1 bash tools/gpu_pmc_sq.sh 2>&1 | grep "dmff\|cross_att\|ceiling" | head -12
cp gpurun_out/pmc_sq_summary.json gpurun_out/pmc_sq_summary_default.json; cp gpurun_out/pmc_sq_insts.json gpurun_out/pmc_sq_insts_default.json
cd $R && bash tools/gpu_bench.sh 2>&1 | grep -v "^\"\|^W2026\|^{" | tail -4
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/icaf_raw/prof_d1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icaf_raw/prof_d1 -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --depth 1 --no-overlap > $R/gpurun_out/prof_bench_depth1.json 2> $R/gpurun_out/prof_depth1.err
f=$(find /tmp/icaf_raw/prof_d1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_depth1_kernel_stats.csv
cd $R && timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -1 gpurun_out/layer_profile.txt
cd $R && ICAF_DMFF_FUSE=0 timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile_plain.txt 2>/dev/null; head -1 gpurun_out/layer_profile_plain.txt
name=c4_s_bf16_b64_512x640_loops3
ARGS="--loops 3 --height 512 --width 640 --batch 64"
cd $R && PMC_NAME=$name bash tools/gpu_pmc.sh --tune-cache $R/profiles/tune_cache_$name.json $ARGS 2>&1 | tail -1
cd $R; timeout 900 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_$name.json $ARGS > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
cd /tmp && rm -rf /tmp/icaf_raw/prof_$name
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icaf_raw/prof_$name -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --tune-cache $R/profiles/tune_cache_$name.json $ARGS > $R/gpurun_out/prof_bench_$name.json 2> $R/gpurun_out/prof_$name.err
f=$(find /tmp/icaf_raw/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_${name}_kernel_stats.csv
cd $R; python - <<'PY'
import json
for n in ("bench.json", "bench_c4_s_bf16_b64_512x640_loops3.json"):
    d = json.load(open("gpurun_out/" + n)); r = d["roofline"]
    print(n, d["value"], d["value_min"], d["value_max"], "fwd", d["forward_ms_per_batch"], d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"], r["kernel"], r["frac"], r["avg_launch_us"], r["traffic"], d["forward_roofline"].get("traffic"), d["forward_roofline"].get("traffic_kernels_without_counters"), (d.get("h2d_feed") or {}).get("pairs_per_s_with_h2d"))
PY
