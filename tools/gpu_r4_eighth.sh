#!/bin/bash
# Round 4, eighth GPU call: igemm_wreg with 64 channels per wave (128 x 512 eight waves, 128 x 256 four waves): kernel tests, bit-identity at full grid,
# then the committed tile choices against the same caches with the new tiles offered (ICAF_RETUNE_TILES: a newcomer must beat the cached choice by 3 %).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "from_registers" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/t8a.log 2>&1; tail -1 gpurun_out/t8a.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t8a.log | sort | uniq -c | head
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider > gpurun_out/t8b.log 2>&1; tail -1 gpurun_out/t8b.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t8b.log | sort | uniq -c | head
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
  timeout 600 python bench.py $B --tune-cache $R/profiles/$cache "$@" > gpurun_out/b8_${name}_old.json 2> gpurun_out/b8_${name}_old.err; q gpurun_out/b8_${name}_old.json
  cp profiles/$cache gpurun_out/tune8_$name.json
  ICAF_RETUNE_TILES=63,64 timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune8_$name.json "$@" > gpurun_out/b8_${name}_new.json 2> gpurun_out/b8_${name}_new.err; q gpurun_out/b8_${name}_new.json
  timeout 600 python bench.py $B --tune-cache $R/profiles/$cache "$@" > gpurun_out/b8_${name}_old2.json 2> gpurun_out/b8_${name}_old2.err; q gpurun_out/b8_${name}_old2.json
  python - "$cache" "$name" <<'PY'
import json, sys
a = {tuple(k): v for k, v in json.load(open(f"profiles/{sys.argv[1]}"))}
b = {tuple(k): v for k, v in json.load(open(f"gpurun_out/tune8_{sys.argv[2]}.json"))}
ch = [(k, a.get(k), v) for k, v in b.items() if a.get(k) != v]
print(f"{sys.argv[2]}: {len(ch)} of {len(b)} signatures changed")
for k, o, n in ch: print(f"   M={k[0]} N={k[1]} Cin={k[2]} k={k[3]} s={k[5]} g={k[11]}: {o} -> {n}")
PY
}
run default tune_cache.json
run c3 tune_cache_c3_l_bf16_b32_640.json --model l --batch 32
run c5 tune_cache_c5_l_vedai_f16_b16_1280.json --model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3
