# fresh tuning runs for the per-GPU shards of BASELINE configs 3 / 4 / 5 (new launch configurations get their chance); the committed caches
# are then updated by hand with the picks of this round's kernels (tile ids > 50) — fresh full tuning is noisy on the small layers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run () { name=$1; shift
  rm -f gpurun_out/tune_new_$name.json
  timeout 600 python bench.py --no-cpu-baseline --no-latency --repeats 3 --tune-cache profiles/tune_cache_$name.json "$@" > gpurun_out/rt_old_$name.json 2>/dev/null
  timeout 900 python bench.py --no-cpu-baseline --no-latency --repeats 3 --tune-cache gpurun_out/tune_new_$name.json "$@" > gpurun_out/rt_new_$name.json 2>/dev/null
  python - $name <<'PY'
import json, sys
n = sys.argv[1]
for t in ("old", "new"):
    d = json.load(open(f"gpurun_out/rt_{t}_{n}.json")); print(n, t, d["value"], d["forward_only_pairs_per_s_one_in_flight"], d["forward_ms_per_batch"])
PY
}
run c3_l_bf16_b32_640 --model l --batch 32
run c4_s_bf16_b64_512x640_loops3 --loops 3 --height 512 --width 640 --batch 64
run c5_l_vedai_f16_b16_1280 --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3
