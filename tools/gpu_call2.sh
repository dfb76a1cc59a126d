#!/bin/bash
# Round 5, second GPU call: the product library with the restructured write-back adopted (ICAF_EPI_FAST), the pre-activation term added in the write-back
# phase (ICAF_PRE_WB) and the persistent GEMM publishing a slice one step early (ICAF_PERS_EARLY, TMAX = 7) — (1) whole GPU suite, (2) ablations of the
# persistent kernel on two yolov5l layers, (3) the fuse convolutions old / new, (4) bench A/B against the call-1 epifast variant (= old pre-term path).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 900 python -m pytest tests -q -m gpu --timeout=300 --tb=short -p no:cacheprovider > gpurun_out/c2_tests.log 2>&1
echo "== GPU suite: $(tail -1 gpurun_out/c2_tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c2_tests.log | head -20
export ICAF_PROBE_NOTUNE=1
L="19:64,67 75:64,67 40:64,67 17:61,67"
ICAF_PROBE_MODEL=l timeout 300 python tools/probes/time_layer.py $L 2>/dev/null | tail -1
for v in pearly0 pabl1 pabl2 pabl4 pabl8 pabl16 pabl3 pabl15; do
  [ -f icafusion_amd/lib/libicaf_$v.so ] && ICAF_LIB=$R/icafusion_amd/lib/libicaf_$v.so ICAF_PROBE_MODEL=l timeout 300 python tools/probes/time_layer.py 19:67 75:67 40:67 17:67 2>/dev/null | tail -1
done
V=icafusion_amd/lib/libicaf_epifast.so
F="26:21,1,28,51 32:28,21,51 39:28,21,51"
timeout 300 python tools/probes/time_layer.py $F 2>/dev/null | tail -1
ICAF_LIB=$R/$V timeout 300 python tools/probes/time_layer.py $F 2>/dev/null | tail -1
unset ICAF_PROBE_NOTUNE
TAG=c2 LIBS="old:$V new:" BENCH="--repeats 5" REPS=2 ICAF_PERS_GEMM=0 FIELDS="kernels.igemm_dma128x2_bf16_bf16_128x128 kernels.igemm_dma128x2_bf16_bf16_128x128w8" bash tools/gpu_ab.sh
cp profiles/tune_cache_c3_l_bf16_b32_640.json /tmp/c3r.json
ICAF_RETUNE_TILES=67 timeout 400 python bench.py --no-cpu-baseline --no-latency --no-h2d --repeats 3 --model l --batch 32 --tune-cache /tmp/c3r.json > gpurun_out/c2_c3_pers.json 2> gpurun_out/c2_c3_pers.err
cp /tmp/c3r.json gpurun_out/c2_tune_c3.json
ICAF_PERS_GEMM=0 timeout 400 python bench.py --no-cpu-baseline --no-latency --no-h2d --repeats 3 --model l --batch 32 --tune-cache profiles/tune_cache_c3_l_bf16_b32_640.json > gpurun_out/c2_c3_nopers.json 2> gpurun_out/c2_c3_nopers.err
python - <<'PY'
import json
for f in ("c2_c3_pers", "c2_c3_nopers"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "one-in-flight", d.get("forward_only_pairs_per_s_one_in_flight"), {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d["kernels"].items() if "pers" in k or "wreg" in k})
    except Exception as e:
        print(f, "NO RESULT", e)
PY
