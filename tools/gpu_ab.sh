#!/bin/bash
# ONE parametrised same-box A/B for a GPU call (replaces the per-call gpu_r3_* / gpu_r4_* scripts of earlier rounds):
#
#   gpurun -- 'TAG=x TESTS="tests/test_gpu_kernels.py::-k conv3x3" LIBS="base:icafusion_amd/lib/libicaf_base.so new:" \
#              BENCH="--repeats 5" REPS=2 FIELDS="kernels.cwide_bf16_8x8n128" bash tools/gpu_ab.sh'
#
#   TAG     prefix of the files written under gpurun_out/ (default ab)
#   TESTS   space-separated "<pytest path>[::-k <expr with _ for spaces>]" groups, each run in its own process FIRST (a faulting kernel poisons only
#           its own HIP context); TESTS_LIB = the library they run on (default: the product library)
#   LIBS    space-separated "<name>:<path to a variant .so, empty = product library>" (tools/build_variant.py writes the variants); ENVS likewise
#           "<name>:<VAR=value,VAR=value>" for A/B switches read from the environment
#   BENCH   bench.py arguments (always with --no-cpu-baseline --no-latency --no-h2d); CONFIG = c3 / c4 / c5 adds that BASELINE configuration's shard
#           arguments and its committed tune cache (a /tmp copy: a re-tune inside the call must not edit the tracked file)
#   REPS    interleaved repetitions (default 2): lib1 lib2 lib1 lib2 ... — box drift hits every leg alike
#   FIELDS  extra dotted paths into the bench line printed per run (kernels.<name> prints that kernel's microseconds per step)
#   LAYERS  =1: tools/layer_profile.py --autotune once per library -> gpurun_out/<TAG>_layers_<name>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
TAG=${TAG:-ab}; REPS=${REPS:-2}; LIBS=${LIBS:-"new:"}; ENVS=${ENVS:-}
for grp in $TESTS; do
  path=${grp%%::*}; expr=""; [ "$grp" != "$path" ] && expr=$(echo "${grp#*::-k }" | tr '_' ' ')
  name=$(echo "$grp" | tr '/: ' '___'); log=gpurun_out/${TAG}_t_$name.log
  ICAF_LIB=${TESTS_LIB:+$R/$TESTS_LIB} timeout ${TEST_TIMEOUT:-900} python -m pytest $path -q -m gpu ${expr:+-k "$expr"} --timeout=300 --tb=short -p no:cacheprovider > $log 2>&1
  echo "== $grp: $(tail -1 $log)"; grep -E "^(FAILED|ERROR)" $log | head -12
done
case "$CONFIG" in
  c3) CARGS="--model l --batch 32"; CACHE=profiles/tune_cache_c3_l_bf16_b32_640.json ;;
  c4) CARGS="--loops 3 --height 512 --width 640 --batch 64"; CACHE=profiles/tune_cache_c4_s_bf16_b64_512x640_loops3.json ;;
  c5) CARGS="--model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3"; CACHE=profiles/tune_cache_c5_l_vedai_f16_b16_1280.json ;;
  *)  CARGS=""; CACHE="" ;;
esac
[ -n "$CACHE" ] && cp $CACHE /tmp/${TAG}_cache.json && CARGS="$CARGS --tune-cache /tmp/${TAG}_cache.json"
show () { python - "$1" $FIELDS <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:                      # a failed run leaves an empty file: say so, keep going
    print(sys.argv[1].split("/")[-1], "NO RESULT", e); sys.exit(0)
out = [sys.argv[1].split("/")[-1], "value", d["value"], d.get("value_min"), d.get("value_max"), "fwd_ms", d.get("forward_ms_per_batch")]
for path in sys.argv[2:]:
    v = d
    for k in path.split("."):
        v = v.get(k) if isinstance(v, dict) else None
    if isinstance(v, dict) and "ms_per_step" in v:
        v = round(v["ms_per_step"] * 1e3, 1)
    out += [path.split(".")[-1], v]
print(*out)
PY
}
if [ -n "${BENCH+x}" ] || [ -n "$CONFIG" ]; then
  for r in $(seq 1 $REPS); do
    for spec in $LIBS ${ENVS:+$ENVS}; do
      name=${spec%%:*}; val=${spec#*:}; f=gpurun_out/${TAG}_${name}$r
      if [[ " $ENVS " == *" $spec "* ]]; then
        env $(echo "$val" | tr ',' ' ') timeout ${BENCH_TIMEOUT:-300} python bench.py --no-cpu-baseline --no-latency --no-h2d $CARGS $BENCH > $f.json 2> $f.err
      else
        ICAF_LIB=${val:+$R/$val} timeout ${BENCH_TIMEOUT:-300} python bench.py --no-cpu-baseline --no-latency --no-h2d $CARGS $BENCH > $f.json 2> $f.err
      fi
      show $f.json
    done
  done
fi
if [ "$LAYERS" = 1 ]; then
  for spec in $LIBS; do
    name=${spec%%:*}; val=${spec#*:}
    ICAF_LIB=${val:+$R/$val} timeout 300 python tools/layer_profile.py --autotune $LAYER_ARGS > gpurun_out/${TAG}_layers_$name.txt 2>/dev/null; head -1 gpurun_out/${TAG}_layers_$name.txt
  done
fi
