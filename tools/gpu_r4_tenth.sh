#!/bin/bash
# Round 4, tenth GPU call: GELU on the four-wave round-4 tiles (kernel tests), the yolov5l caches with tile 64 offered to the GELU layers too.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "from_registers" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/t10a.log 2>&1; tail -1 gpurun_out/t10a.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t10a.log | sort | uniq -c | head
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
run () {   # name, cache, bench args
  name=$1; cache=$2; shift 2
  timeout 600 python bench.py $B --tune-cache $R/profiles/$cache "$@" > gpurun_out/b10_${name}_old.json 2> gpurun_out/b10_${name}_old.err; q gpurun_out/b10_${name}_old.json
  cp profiles/$cache gpurun_out/tune10_$name.json
  ICAF_RETUNE_TILES=64 timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune10_$name.json "$@" > gpurun_out/b10_${name}_new.json 2> gpurun_out/b10_${name}_new.err; q gpurun_out/b10_${name}_new.json
  timeout 600 python bench.py $B --tune-cache $R/profiles/$cache "$@" > gpurun_out/b10_${name}_old2.json 2> gpurun_out/b10_${name}_old2.err; q gpurun_out/b10_${name}_old2.json
  timeout 600 python bench.py $B --tune-cache $R/gpurun_out/tune10_$name.json "$@" > gpurun_out/b10_${name}_new2.json 2> gpurun_out/b10_${name}_new2.err; q gpurun_out/b10_${name}_new2.json
  python - "$cache" "$name" <<'PY'
import json, sys
a = {tuple(k): v for k, v in json.load(open(f"profiles/{sys.argv[1]}"))}
b = {tuple(k): v for k, v in json.load(open(f"gpurun_out/tune10_{sys.argv[2]}.json"))}
ch = [(k, a.get(k), v) for k, v in b.items() if a.get(k) != v]
print(f"{sys.argv[2]}: {len(ch)} of {len(b)} signatures changed")
for k, o, n in ch: print(f"   M={k[0]} N={k[1]} Cin={k[2]} k={k[3]} s={k[5]} g={k[11]} act={k[14]}: {o} -> {n}")
PY
}
run c3 tune_cache_c3_l_bf16_b32_640.json --model l --batch 32
run c5 tune_cache_c5_l_vedai_f16_b16_1280.json --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3
