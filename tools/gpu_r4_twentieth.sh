#!/bin/bash
# Round 4, twentieth GPU call: icaf_bottleneck + cv3 with the checked fast write-back (the library) against the shared epilogue's general loop
# (libicaf_oldbneck2.so): kernel tests, plan-level bit-identity, the kernel's time and the bench, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "bottleneck" --timeout=120 --tb=short -p no:cacheprovider > gpurun_out/t20.log 2>&1
echo "== bottleneck tests: $(tail -1 gpurun_out/t20.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t20.log | head
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "bneck us", round(d["kernels"]["bottleneck+cv3"]["ms_per_step"] * 1e3, 1))
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 5"
for r in 1 2; do
  ICAF_LIB=$R/icafusion_amd/lib/libicaf_oldbneck2.so timeout 300 python bench.py $B > gpurun_out/b20_old$r.json 2> gpurun_out/b20_old$r.err; q gpurun_out/b20_old$r.json
  timeout 300 python bench.py $B > gpurun_out/b20_new$r.json 2> gpurun_out/b20_new$r.err; q gpurun_out/b20_new$r.json
done
