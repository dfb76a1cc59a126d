#!/bin/bash
# Round 4, twenty-first GPU call: igemm_wreg's 64-pixel / 64-channel tiles with the checked fast write-back (the library) against the general loop only
# (libicaf_oldwreg.so): the kernel's own tests, the full-grid bit-identity of every launch configuration, the bench, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -m gpu -k "from_registers or bit_identical" --timeout=120 --tb=short -p no:cacheprovider > gpurun_out/t21.log 2>&1
echo "== wreg + full-grid tests: $(tail -1 gpurun_out/t21.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/t21.log | head
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d["kernels"].items() if k.startswith("igemm_wreg")})
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 5"
for r in 1 2; do
  ICAF_LIB=$R/icafusion_amd/lib/libicaf_oldwreg.so timeout 200 python bench.py $B > gpurun_out/b21_old$r.json 2> gpurun_out/b21_old$r.err; q gpurun_out/b21_old$r.json
  timeout 200 python bench.py $B > gpurun_out/b21_new$r.json 2> gpurun_out/b21_new$r.err; q gpurun_out/b21_new$r.json
done
