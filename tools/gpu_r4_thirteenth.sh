#!/bin/bash
# Round 4, thirteenth GPU call: the C3 tail on the other BASELINE shards (yolov5l: the tail sits at 80 x 80 / 160 x 160 with 4-16 x the tiles),
# and the 8 x 8 form forced on the default workload.  Same-box A/B, switch off / on, the new signatures tuned into copies of the committed caches.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 5"
ab () {  # name, args
  name=$1; shift
  cp profiles/tune_cache_$name.json gpurun_out/tune13_$name.json
  ICAF_C3_TAIL=0 timeout 600 python bench.py $B --tune-cache $R/profiles/tune_cache_$name.json "$@" > gpurun_out/b13_${name}_off.json 2> gpurun_out/b13_${name}_off.err; q gpurun_out/b13_${name}_off.json
  timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune13_$name.json "$@" > gpurun_out/b13_${name}_on.json 2> gpurun_out/b13_${name}_on.err; q gpurun_out/b13_${name}_on.json
  ICAF_C3_TAIL=0 timeout 600 python bench.py $B --tune-cache $R/profiles/tune_cache_$name.json "$@" > gpurun_out/b13_${name}_off2.json 2> gpurun_out/b13_${name}_off2.err; q gpurun_out/b13_${name}_off2.json
  timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune13_$name.json "$@" > gpurun_out/b13_${name}_on2.json 2> gpurun_out/b13_${name}_on2.err; q gpurun_out/b13_${name}_on2.json
}
git_cached () { cp profiles/tune_cache_$1.json /tmp/keep_$1.json; }
for n in c3_l_bf16_b32_640 c4_s_bf16_b64_512x640_loops3 c5_l_vedai_f16_b16_1280; do git_cached $n; done
ab c3_l_bf16_b32_640 --model l --batch 32
ab c5_l_vedai_f16_b16_1280 --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3
ab c4_s_bf16_b64_512x640_loops3 --loops 3 --height 512 --width 640 --batch 64
for n in c3_l_bf16_b32_640 c4_s_bf16_b64_512x640_loops3 c5_l_vedai_f16_b16_1280; do cp /tmp/keep_$n.json profiles/tune_cache_$n.json; done    # (the OFF runs must not have rewritten the committed files)
# default workload with the 8 x 8 form (3 workgroups per CU, 168 registers) forced on the tails
python - <<'PY'
import json
c = json.load(open("tune12_tmp.json"))          # (the cache the twelfth call tuned: the committed one + the two tail signatures)
out = [[k, (82 if k[18] == 2 else v)] for k, v in c]
json.dump(out, open("gpurun_out/tune13_default82.json", "w"))
print("tails in cache:", [(k[0], k[11], v) for k, v in out if k[18] == 2])
PY
timeout 600 python bench.py $B --tune-cache $R/gpurun_out/tune13_default82.json > gpurun_out/b13_default82.json 2> gpurun_out/b13_default82.err; q gpurun_out/b13_default82.json
ICAF_C3_TAIL=0 timeout 600 python bench.py $B > gpurun_out/b13_default_off.json 2> gpurun_out/b13_default_off.err; q gpurun_out/b13_default_off.json
python - <<'PY'
import json
c = json.load(open("gpurun_out/tune13_default82.json"))
print("after the run:", [(k[0], k[11], k[9], k[10], v) for k, v in c if len(k) > 18 and k[18] == 2])
PY
