#!/bin/bash
# Round-4 evidence, one GPU call (raw profiler output stays under /tmp on the box): for the default workload and the per-GPU shards of BASELINE
# configs 3 / 4 / 5 the two PMC traffic passes (installed as profiles/pmc_traffic*.json so that the bench line that follows carries
# roofline.traffic AND forward_roofline.traffic), the bench line, rocprofv3 kernel stats of the same command; for the default workload also a
# DEPTH-1 kernel trace (one batch at a time: its per-kernel averages are what roofline.avg_launch_us reports; the default trace is of the
# overlapped run), the SQ passes (MFMA busy + instruction mix) for the default workload and the yolov5l shard, the per-layer profiles and the
# 16-bit parity table (with the small-object mAP recipe).  Everything lands in gpurun_out/; summaries are copied to profiles/r04_* afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +     # the snapshot's old scratch: gpurun_out/ is capped at 64 MiB
cd $R && bash tools/gpu_pmc.sh 2>&1 | tail -2
cd $R && SQ_INSTS=1 bash tools/gpu_pmc_sq.sh 2>&1 | grep "dmff\|cross_att\|stem\|ceiling" | head -40
cp gpurun_out/pmc_sq_summary.json gpurun_out/pmc_sq_summary_default.json; cp gpurun_out/pmc_sq_insts.json gpurun_out/pmc_sq_insts_default.json
cd $R && bash tools/gpu_bench.sh 2>&1 | grep -v "^\"\|^W2026" | tail -4
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/icaf_raw/prof_d1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icaf_raw/prof_d1 -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --depth 1 --no-overlap > $R/gpurun_out/prof_bench_depth1.json 2> $R/gpurun_out/prof_depth1.err
f=$(find /tmp/icaf_raw/prof_d1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_depth1_kernel_stats.csv && cut -c1-140 "$f" | sed -n 2,4p
cd $R && timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -1 gpurun_out/layer_profile.txt
cd $R && ICAF_DMFF_FUSE=0 timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile_plain.txt 2>/dev/null; head -1 gpurun_out/layer_profile_plain.txt
cd $R && timeout 1200 python tools/parity16.py --out gpurun_out/parity_16bit.json > gpurun_out/parity16.log 2>&1; tail -1 gpurun_out/parity16.log | cut -c1-200
run_cfg () {   # name, bench args...
  name=$1; shift
  cd $R && PMC_NAME=$name bash tools/gpu_pmc.sh --tune-cache $R/profiles/tune_cache_$name.json "$@" 2>&1 | tail -1
  cd $R
  timeout 900 python bench.py --no-cpu-baseline --tune-cache $R/profiles/tune_cache_$name.json "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/bench_{n}.json"))
    print(n, {k: d[k] for k in ("value", "value_min", "value_max", "ms_per_step", "forward_only_pairs_per_s", "nms_ms_per_batch_standalone", "forward_roofline")}, d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["algorithmic_bytes_per_launch"])
except Exception as e:
    print(n, "FAILED", e); print(open(f"gpurun_out/bench_{n}.err").read()[-1500:])
PY
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/icaf_raw/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icaf_raw/prof_$name -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --tune-cache $R/profiles/tune_cache_$name.json "$@" > $R/gpurun_out/prof_bench_$name.json 2> $R/gpurun_out/prof_$name.err
  f=$(find /tmp/icaf_raw/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_${name}_kernel_stats.csv && cut -c1-140 "$f" | sed -n 2,3p
}
run_cfg c3_l_bf16_b32_640 --model l --batch 32
cd $R && SQ_INSTS=1 bash tools/gpu_pmc_sq.sh --model l --batch 32 --tune-cache $R/profiles/tune_cache_c3_l_bf16_b32_640.json > gpurun_out/pmc_sq_c3.log 2>&1
cp gpurun_out/pmc_sq_summary.json gpurun_out/pmc_sq_summary_c3.json; cp gpurun_out/pmc_sq_insts.json gpurun_out/pmc_sq_insts_c3.json; grep "dmff\|cross_att" gpurun_out/pmc_sq_c3.log
run_cfg c4_s_bf16_b64_512x640_loops3 --loops 3 --height 512 --width 640 --batch 64
run_cfg c5_l_vedai_f16_b16_1280 --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3
