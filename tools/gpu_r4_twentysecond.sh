#!/bin/bash
# Round 4, twenty-second GPU call: the same A/B (igemm_wreg fast write-back, libicaf_oldwreg.so = before) on the yolov5l shard, whose dominant kernel is the 128x256w4 tile.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d["kernels"].items() if k.startswith("igemm_wreg")})
PY
}
cp profiles/tune_cache_c3_l_bf16_b32_640.json /tmp/c3.json
B="--no-cpu-baseline --no-latency --no-h2d --repeats 3 --model l --batch 32 --tune-cache /tmp/c3.json"
for r in 1 2; do
  ICAF_LIB=$R/icafusion_amd/lib/libicaf_oldwreg.so timeout 100 python bench.py $B > gpurun_out/b22_old$r.json 2> gpurun_out/b22_old$r.err; q gpurun_out/b22_old$r.json
  timeout 100 python bench.py $B > gpurun_out/b22_new$r.json 2> gpurun_out/b22_new$r.err; q gpurun_out/b22_new$r.json
done
