#!/bin/bash
# Round 4, fourth GPU call: throughput A/Bs on one box (batches in flight 2 / 3; P3 DMFF as two launches or three), then the SQ counter passes on
# the yolov5l shard (config 3: DMFF at d_k = 32 / 64 / 128).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
q () { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"], "mfma", d["forward_roofline"]["mfma_frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
timeout 300 python bench.py $B > gpurun_out/b_default.json 2> gpurun_out/b_default.err; q gpurun_out/b_default.json
ICAF_DMFF_FUSE_MAX_C=64 timeout 300 python bench.py $B > gpurun_out/b_p3three.json 2> gpurun_out/b_p3three.err; q gpurun_out/b_p3three.json
timeout 300 python bench.py $B --depth 3 > gpurun_out/b_depth3.json 2> gpurun_out/b_depth3.err; q gpurun_out/b_depth3.json
ICAF_DMFF_FUSE_MAX_C=64 timeout 300 python bench.py $B --depth 3 > gpurun_out/b_p3three_depth3.json 2> gpurun_out/b_p3three_depth3.err; q gpurun_out/b_p3three_depth3.json
timeout 300 python bench.py $B > gpurun_out/b_default2.json 2> gpurun_out/b_default2.err; q gpurun_out/b_default2.json
echo "== c3 shard (yolov5l bf16 b32): bench + SQ passes"
timeout 600 python bench.py $B --model l --batch 32 --tune-cache $R/profiles/tune_cache_c3_l_bf16_b32_640.json > gpurun_out/b_c3.json 2> gpurun_out/b_c3.err; q gpurun_out/b_c3.json
SQ_INSTS=1 bash tools/gpu_pmc_sq.sh --model l --batch 32 --tune-cache $R/profiles/tune_cache_c3_l_bf16_b32_640.json > gpurun_out/pmc_sq_c3.log 2>&1
cp gpurun_out/pmc_sq_summary.json gpurun_out/pmc_sq_summary_c3.json; cp gpurun_out/pmc_sq_insts.json gpurun_out/pmc_sq_insts_c3.json
grep "dmff\|cross_att\|layernorm" gpurun_out/pmc_sq_c3.log
