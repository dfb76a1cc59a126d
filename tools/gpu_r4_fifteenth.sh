#!/bin/bash
# Round 4, fifteenth GPU call: icaf_bottleneck + cv3 with the residual / cv2 vectors requested beside the patch (the library) against the form that asks
# for them behind the 3x3 loop (libicaf_late3.so): bit-identity tests of the kernel, its layer-profile row, PMC bytes, and the whole bench, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bottleneck" --timeout=180 --tb=short -p no:cacheprovider > gpurun_out/t15_kernels.log 2>&1
echo "== kernels: $(tail -1 gpurun_out/t15_kernels.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t15_kernels.log | head
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "bottleneck_cv3 or stem2_plan" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/t15_model.log 2>&1
echo "== model: $(tail -1 gpurun_out/t15_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t15_model.log | head
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
for r in 1 2 3; do
  ICAF_LIB=$R/icafusion_amd/lib/libicaf_late3.so timeout 600 python bench.py $B > gpurun_out/b15_late$r.json 2> gpurun_out/b15_late$r.err; q gpurun_out/b15_late$r.json
  timeout 600 python bench.py $B > gpurun_out/b15_early$r.json 2> gpurun_out/b15_early$r.err; q gpurun_out/b15_early$r.json
done
ICAF_LIB=$R/icafusion_amd/lib/libicaf_late3.so timeout 300 python tools/layer_profile.py --autotune 2>/dev/null | sed -n 1,4p
timeout 300 python tools/layer_profile.py --autotune 2>/dev/null | sed -n 1,4p
