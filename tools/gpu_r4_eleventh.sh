#!/bin/bash
# Round 4, eleventh GPU call: the 64-pixel wreg tiles offered to launches of up to 512 k pixels (default workload), same-box A/B against the committed cache.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
timeout 600 python bench.py $B > gpurun_out/b11_old.json 2> gpurun_out/b11_old.err; q gpurun_out/b11_old.json
cp profiles/tune_cache.json gpurun_out/tune11.json
ICAF_WREG64_MAXPIX=524288 ICAF_RETUNE_TILES=65,66 timeout 900 python bench.py $B --tune-cache $R/gpurun_out/tune11.json > gpurun_out/b11_new.json 2> gpurun_out/b11_new.err; q gpurun_out/b11_new.json
timeout 600 python bench.py $B > gpurun_out/b11_old2.json 2> gpurun_out/b11_old2.err; q gpurun_out/b11_old2.json
ICAF_WREG64_MAXPIX=524288 timeout 600 python bench.py $B --tune-cache $R/gpurun_out/tune11.json > gpurun_out/b11_new2.json 2> gpurun_out/b11_new2.err; q gpurun_out/b11_new2.json
python - <<'PY'
import json
a = {tuple(k): v for k, v in json.load(open("profiles/tune_cache.json"))}
b = {tuple(k): v for k, v in json.load(open("gpurun_out/tune11.json"))}
ch = [(k, a.get(k), v) for k, v in b.items() if a.get(k) != v]
print(f"{len(ch)} of {len(b)} signatures changed")
for k, o, n in ch: print(f"   M={k[0]} N={k[1]} Cin={k[2]} k={k[3]} s={k[5]} g={k[11]}: {o} -> {n}")
PY
