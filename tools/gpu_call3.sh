#!/bin/bash
# Round 5, third GPU call: (1) the serving loop fed from pinned host memory with ONE copy per batch (depth + 1 plans), (2) attention: query splits per
# head (K / V^T staging amortised over 1 / 2 / 4 query tiles per wave), (3) the VEDAI shard with launch configuration 67 offered.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/c3_pipe.log 2>&1
echo "== pipeline tests: $(tail -1 gpurun_out/c3_pipe.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c3_pipe.log | head
ICAF_PERS_GEMM=0 timeout 400 python bench.py --no-cpu-baseline --no-latency --repeats 5 > gpurun_out/c3_h2d.json 2> gpurun_out/c3_h2d.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/c3_h2d.json")); print("h2d", d["value"], d["forward_ms_per_batch"], json.dumps(d.get("h2d_feed"))[:600])
except Exception as e:
    print("h2d NO RESULT", e)
PY
for m in s l; do
  for qs in 0 1 2 3; do
    echo "model $m qsplit $qs"; ICAF_ATTN_QSPLIT=$qs timeout 200 python tools/probes/dmff_levels.py $m 2>/dev/null | grep three | sed 's/.*| //'
  done
done
cp profiles/tune_cache_c5_l_vedai_f16_b16_1280.json /tmp/c5r.json
A5="--no-cpu-baseline --no-latency --no-h2d --repeats 3 --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3"
ICAF_RETUNE_TILES=67 timeout 500 python bench.py $A5 --tune-cache /tmp/c5r.json > gpurun_out/c3_c5_pers.json 2> gpurun_out/c3_c5_pers.err
cp /tmp/c5r.json gpurun_out/c3_tune_c5.json
ICAF_PERS_GEMM=0 timeout 500 python bench.py $A5 --tune-cache profiles/tune_cache_c5_l_vedai_f16_b16_1280.json > gpurun_out/c3_c5_nopers.json 2> gpurun_out/c3_c5_nopers.err
python - <<'PY'
import json
for f in ("c3_c5_pers", "c3_c5_nopers"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "one-in-flight", d.get("forward_only_pairs_per_s_one_in_flight"), {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d["kernels"].items() if "pers" in k or "wreg" in k})
    except Exception as e:
        print(f, "NO RESULT", e)
PY
