#!/bin/bash
# Round 5, seventh GPU call: the fuse convolutions (pre-activation term) re-tuned in every committed cache — the igemm tiles add the term in the write-back phase
# now (coalesced taps), the streaming kernel (51 / 52) still the old way — and the whole bench with the re-tuned cache against the committed one, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
B="--no-cpu-baseline --no-latency --no-h2d --repeats 5"
run () {   # tag, cache name (in profiles/), bench args
  tag=$1; cache=$2; shift 2
  cp profiles/$cache /tmp/rt_$tag.json
  ICAF_RETUNE_PRE=1 timeout 500 python bench.py $B --tune-cache /tmp/rt_$tag.json "$@" > gpurun_out/c7_${tag}_new.json 2> gpurun_out/c7_${tag}_new.err
  cp /tmp/rt_$tag.json gpurun_out/c7_tune_$tag.json
  timeout 500 python bench.py $B --tune-cache profiles/$cache "$@" > gpurun_out/c7_${tag}_old.json 2> gpurun_out/c7_${tag}_old.err
  timeout 500 python bench.py $B --tune-cache gpurun_out/c7_tune_$tag.json "$@" > gpurun_out/c7_${tag}_new2.json 2> gpurun_out/c7_${tag}_new2.err
  python - $tag $cache <<'PY'
import json, sys
tag, cache = sys.argv[1:3]
for leg in ("new", "old", "new2"):
    try:
        d = json.load(open(f"gpurun_out/c7_{tag}_{leg}.json"))
        print(tag, leg, "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"])
    except Exception as e:
        print(tag, leg, "NO RESULT", e)
a = {tuple(k): t for k, t in json.load(open(f"profiles/{cache}"))}
b = {tuple(k): t for k, t in json.load(open(f"gpurun_out/c7_tune_{tag}.json"))}
print(tag, "changed:", [(k[0], k[1], k[2], a.get(k), t) for k, t in b.items() if a.get(k) != t])
PY
}
run default tune_cache.json
run c4 tune_cache_c4_s_bf16_b64_512x640_loops3.json --loops 3 --height 512 --width 640 --batch 64
run c3 tune_cache_c3_l_bf16_b32_640.json --model l --batch 32
B="--no-cpu-baseline --no-latency --no-h2d --repeats 3"
run c5 tune_cache_c5_l_vedai_f16_b16_1280.json --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3
