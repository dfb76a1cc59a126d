#!/bin/bash
# Round 4, twelfth GPU call: the C3 tail (cwide.hip CHAIN = 2: a C3's cv3 on its last Bottleneck's 3x3) — parity tests, then a same-box A/B
# of the default workload with the switch off / on (the new signatures are tuned into a copy of the committed cache), then the layer profile.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "c3_tail or resident_patch_streamed or chained or halo_patch" --timeout=180 --tb=short -p no:cacheprovider > gpurun_out/t12_kernels.log 2>&1
echo "== kernels: $(tail -1 gpurun_out/t12_kernels.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t12_kernels.log | head -20; grep -E "AssertionError|Error:|assert " gpurun_out/t12_kernels.log | sort | uniq -c | sort -rn | head -12
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -m gpu -k "c3_tail or chained_bottlenecks or bit_identical or shard" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/t12_model.log 2>&1
echo "== model: $(tail -1 gpurun_out/t12_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t12_model.log | head -20; grep -E "AssertionError|Error:|assert " gpurun_out/t12_model.log | sort | uniq -c | sort -rn | head -12
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
cp profiles/tune_cache.json gpurun_out/tune12.json
ICAF_C3_TAIL=0 timeout 600 python bench.py $B > gpurun_out/b12_off.json 2> gpurun_out/b12_off.err; q gpurun_out/b12_off.json
timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune12.json > gpurun_out/b12_on.json 2> gpurun_out/b12_on.err; q gpurun_out/b12_on.json
ICAF_C3_TAIL=0 timeout 600 python bench.py $B > gpurun_out/b12_off2.json 2> gpurun_out/b12_off2.err; q gpurun_out/b12_off2.json
timeout 600 python bench.py $B --tune-cache $R/gpurun_out/tune12.json > gpurun_out/b12_on2.json 2> gpurun_out/b12_on2.err; q gpurun_out/b12_on2.json
tail -3 gpurun_out/b12_on.err
python - <<'PY'
import json
a = {tuple(k): v for k, v in json.load(open("profiles/tune_cache.json"))}
b = {tuple(k): v for k, v in json.load(open("gpurun_out/tune12.json"))}
for k, v in b.items():
    if a.get(k) != v: print(f"   M={k[0]} N={k[1]} Cin={k[2]} k={k[3]} s={k[5]} g={k[11]} chain={k[17]},{k[18]}: {a.get(k)} -> {v}")
PY
timeout 600 python tools/layer_profile.py --tune-cache $R/gpurun_out/tune12.json > gpurun_out/layer_profile_tail.txt 2> gpurun_out/layer_profile_tail.err; grep -n "cv3\|total" gpurun_out/layer_profile_tail.txt | head; tail -2 gpurun_out/layer_profile_tail.err
