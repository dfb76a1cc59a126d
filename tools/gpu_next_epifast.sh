#!/bin/bash
# Prepared at the end of round 4, NOT yet run: the shared epilogue's restructured write-back (conv_common.h, -DICAF_EPI_FAST=1: two checked fast loops + one
# compact general loop) as a variant library against the product library.
#   python tools/build_variant.py epifast -DICAF_EPI_FAST=1        (3.5 min; compiles clean: 0 kernels with scratch, occupancy up in 22 instantiations, down in 3 rare ones)
#   gpurun -- 'bash tools/gpu_next_epifast.sh'
# Step 1: the whole GPU suite on the variant (it replaces the write-back of igemm_dma / igemm_wreg / ctile / stem kernels: every bit-identity test applies).
# Step 2: same-box A/B of the bench, default workload and the yolov5l shard.  Adopt = flip the default of ICAF_EPI_FAST and drop igemm_wreg's own FAST_WB test.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
V=$R/icafusion_amd/lib/libicaf_epifast.so
[ -f $V ] || { echo "build the variant first: python tools/build_variant.py epifast -DICAF_EPI_FAST=1"; exit 1; }
ICAF_LIB=$V timeout 1200 python -m pytest tests -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -x > gpurun_out/epifast_tests.log 2>&1
echo "== GPU suite on the variant: $(tail -1 gpurun_out/epifast_tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/epifast_tests.log | head
q () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"])
PY
}
B="--no-cpu-baseline --no-latency --no-h2d --repeats 7"
for r in 1 2 3; do
  timeout 300 python bench.py $B > gpurun_out/epi_old$r.json 2> gpurun_out/epi_old$r.err; q gpurun_out/epi_old$r.json
  ICAF_LIB=$V timeout 300 python bench.py $B > gpurun_out/epi_new$r.json 2> gpurun_out/epi_new$r.err; q gpurun_out/epi_new$r.json
done
cp profiles/tune_cache_c3_l_bf16_b32_640.json /tmp/c3.json
L="--no-cpu-baseline --no-latency --no-h2d --repeats 3 --model l --batch 32 --tune-cache /tmp/c3.json"
for r in 1 2; do
  timeout 300 python bench.py $L > gpurun_out/epi_c3_old$r.json 2> gpurun_out/epi_c3_old$r.err; q gpurun_out/epi_c3_old$r.json
  ICAF_LIB=$V timeout 300 python bench.py $L > gpurun_out/epi_c3_new$r.json 2> gpurun_out/epi_c3_new$r.err; q gpurun_out/epi_c3_new$r.json
done
