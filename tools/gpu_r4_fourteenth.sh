#!/bin/bash
# Round 4, fourteenth GPU call: the C3 tail with the candidate rule (8 x 8 form only below 200 k pixels) on the two yolov5s workloads, same-box A/B off / on.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
ab () {  # name, cache, args
  name=$1; cache=$2; shift; shift
  cp $cache /tmp/keep_$name.json; cp $cache gpurun_out/tune14_$name.json
  for r in 1 2 3; do
    ICAF_C3_TAIL=0 timeout 600 python bench.py $B --tune-cache /tmp/keep_$name.json "$@" > gpurun_out/b14_${name}_off$r.json 2> gpurun_out/b14_${name}_off$r.err; q gpurun_out/b14_${name}_off$r.json
    timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune14_$name.json "$@" > gpurun_out/b14_${name}_on$r.json 2> gpurun_out/b14_${name}_on$r.err; q gpurun_out/b14_${name}_on$r.json
  done
  python - "$name" <<'PY'
import json, sys
c = json.load(open(f"gpurun_out/tune14_{sys.argv[1]}.json"))
print("tails:", [(k[0], k[11], v) for k, v in c if k[18] == 2])
PY
}
ab default profiles/tune_cache.json
ab c4 profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json --loops 3 --height 512 --width 640 --batch 64
