#!/bin/bash
# Round 4, ninth GPU call: igemm_wreg with 64-pixel tiles (64 x 256, 64 x 128): kernel tests, bit-identity at full grid, then the committed caches against the
# same caches with the new tiles offered.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "from_registers" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/t9a.log 2>&1; tail -1 gpurun_out/t9a.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t9a.log | sort | uniq -c | head
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider > gpurun_out/t9b.log 2>&1; tail -1 gpurun_out/t9b.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t9b.log | sort | uniq -c | head
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
  timeout 600 python bench.py $B --tune-cache $R/profiles/$cache "$@" > gpurun_out/b9_${name}_old.json 2> gpurun_out/b9_${name}_old.err; q gpurun_out/b9_${name}_old.json
  cp profiles/$cache gpurun_out/tune9_$name.json
  ICAF_RETUNE_TILES=65,66 timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune9_$name.json "$@" > gpurun_out/b9_${name}_new.json 2> gpurun_out/b9_${name}_new.err; q gpurun_out/b9_${name}_new.json
  timeout 600 python bench.py $B --tune-cache $R/profiles/$cache "$@" > gpurun_out/b9_${name}_old2.json 2> gpurun_out/b9_${name}_old2.err; q gpurun_out/b9_${name}_old2.json
  timeout 600 python bench.py $B --tune-cache $R/gpurun_out/tune9_$name.json "$@" > gpurun_out/b9_${name}_new2.json 2> gpurun_out/b9_${name}_new2.err; q gpurun_out/b9_${name}_new2.json
  python - "$cache" "$name" <<'PY'
import json, sys
a = {tuple(k): v for k, v in json.load(open(f"profiles/{sys.argv[1]}"))}
b = {tuple(k): v for k, v in json.load(open(f"gpurun_out/tune9_{sys.argv[2]}.json"))}
ch = [(k, a.get(k), v) for k, v in b.items() if a.get(k) != v]
print(f"{sys.argv[2]}: {len(ch)} of {len(b)} signatures changed")
for k, o, n in ch: print(f"   M={k[0]} N={k[1]} Cin={k[2]} k={k[3]} s={k[5]} g={k[11]}: {o} -> {n}")
PY
}
run default tune_cache.json
run c3 tune_cache_c3_l_bf16_b32_640.json --model l --batch 32
