#!/bin/bash
# Round 4, eighteenth GPU call: icaf_bottleneck with compile-time channel geometry and an unrolled K loop (the library) against the form before
# (libicaf_oldbneck.so): kernel tests, plan-level bit-identity, the kernel's time and the whole bench, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "bottleneck or halo_patch or conv3x3" --timeout=180 --tb=short -p no:cacheprovider > gpurun_out/t18_kernels.log 2>&1
echo "== kernels: $(tail -1 gpurun_out/t18_kernels.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t18_kernels.log | head
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -m gpu -k "bottleneck or golden or shard or bit_identical" --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/t18_model.log 2>&1
echo "== model: $(tail -1 gpurun_out/t18_model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t18_model.log | head
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"], "bneck us", d["kernels"]["bottleneck+cv3"]["ms_per_step"] * 1e3)
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
for r in 1 2 3; do
  ICAF_LIB=$R/icafusion_amd/lib/libicaf_oldbneck.so timeout 600 python bench.py $B > gpurun_out/b18_old$r.json 2> gpurun_out/b18_old$r.err; q gpurun_out/b18_old$r.json
  timeout 600 python bench.py $B > gpurun_out/b18_new$r.json 2> gpurun_out/b18_new$r.err; q gpurun_out/b18_new$r.json
done
