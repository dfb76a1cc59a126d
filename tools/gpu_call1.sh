#!/bin/bash
# Round 5, first GPU call: (1) the new persistent long-K GEMM (launch configuration 67) — kernel tests, then its time on the yolov5l shard's layers against the
# round-4 choices; (2) the stem2 halo fix; (3) the ICAF_EPI_FAST write-back as a variant library: whole GPU suite on it, then same-box bench A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "persistent or stem2" --timeout=200 --tb=short -p no:cacheprovider > gpurun_out/c1_pers.log 2>&1
echo "== persistent + stem2 tests: $(tail -1 gpurun_out/c1_pers.log)"; grep -E "^(FAILED|ERROR)|Error|differ|diff " gpurun_out/c1_pers.log | head -30
export ICAF_PROBE_MODEL=l
L="7:64,67 16:64,67 19:64,62,67 37:64,67 40:64,63,67 75:64,67 88:64,67 97:62,67 100:62,67 17:11,61,67 36:51,61,67 38:61,67 45:61,67 46:64,67 48:61,67 60:51,61,67 70:28,61,67 73:61,67 63:62,67 67:64,67 68:62,67 98:64,67"
timeout 600 python tools/probes/time_layer.py $L 2> gpurun_out/c1_time_l.err | tail -1 | tee gpurun_out/c1_time_l.txt
unset ICAF_PROBE_MODEL
L="7:65,67 13:64,67 14:66,67 16:62,67 17:66,67 20:64,67 39:28,67 42:66,67 55:51,64,67 56:66,67 58:51,64,67 59:66,67"
timeout 600 python tools/probes/time_layer.py $L 2> gpurun_out/c1_time_s.err | tail -1 | tee gpurun_out/c1_time_s.txt
# the whole yolov5l shard with configuration 67 offered against the committed choices (3 % margin), and the default workload likewise
cp profiles/tune_cache_c3_l_bf16_b32_640.json /tmp/c3r.json
ICAF_RETUNE_TILES=67 timeout 400 python bench.py --no-cpu-baseline --no-latency --no-h2d --repeats 3 --model l --batch 32 --tune-cache /tmp/c3r.json > gpurun_out/c1_c3_pers.json 2> gpurun_out/c1_c3_pers.err
cp /tmp/c3r.json gpurun_out/c1_tune_c3.json
cp profiles/tune_cache.json /tmp/dr.json
ICAF_RETUNE_TILES=67 timeout 400 python bench.py --no-cpu-baseline --no-latency --no-h2d --repeats 5 --tune-cache /tmp/dr.json > gpurun_out/c1_def_pers.json 2> gpurun_out/c1_def_pers.err
cp /tmp/dr.json gpurun_out/c1_tune_default.json
python - <<'PY'
import json
for f in ("c1_c3_pers", "c1_def_pers"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], {k: round(v["ms_per_step"] * 1e3, 1) for k, v in d["kernels"].items() if "pers" in k or "wreg" in k})
    except Exception as e:
        print(f, "NO RESULT", e)
PY
V=icafusion_amd/lib/libicaf_epifast.so
if [ -f $V ]; then
  ICAF_LIB=$R/$V timeout 900 python -m pytest tests -q -m gpu --timeout=300 --tb=short -p no:cacheprovider -x > gpurun_out/c1_epifast_tests.log 2>&1
  echo "== GPU suite on the epifast variant: $(tail -1 gpurun_out/c1_epifast_tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/c1_epifast_tests.log | head
  TAG=c1epi LIBS="old: new:$V" BENCH="--repeats 5" REPS=2 ICAF_PERS_GEMM=0 bash tools/gpu_ab.sh
  TAG=c1epi3 LIBS="old: new:$V" CONFIG=c3 BENCH="--repeats 3" REPS=1 ICAF_PERS_GEMM=0 bash tools/gpu_ab.sh
fi
