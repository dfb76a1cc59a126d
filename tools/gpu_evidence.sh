#!/bin/bash
# A round's evidence in ONE GPU call (raw profiler output stays under /tmp on the box; summaries land in gpurun_out/ and are copied to profiles/rNN_* by hand):
#
#   gpurun --timeout 2400 -- 'WORKLOADS="default c3 c4 c5" PARITY=1 bash tools/gpu_evidence.sh'
#
# per workload (default = BASELINE configs[1]; c3 / c4 / c5 = the per-GPU shards of the other GPU configurations, each with its committed tune cache):
#   1. the two PMC traffic passes (tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate runs, kernel-trace only), installed as profiles/pmc_traffic*.json in
#      the box's copy so that the bench line that follows carries roofline.traffic AND forward_roofline.traffic;
#   2. the bench line (with cpu_baseline and h2d_feed for the default workload);
#   3. rocprofv3 --kernel-trace --stats of the same command; for the default workload also a DEPTH-1 trace (one batch at a time: its per-kernel averages are what
#      roofline.avg_launch_us reports — the default trace is of the overlapped run);
#   4. SQ=1 (default): the SQ passes (MFMA busy, VALU issue, wave-parked fractions, instruction mix: tools/gpu_pmc_sq.sh) BEFORE the bench line, installed as
#      profiles/pmc_sq*.json like the traffic passes, so that the line's roofline names the roof that binds from counters of the same call;
#   5. default only: the per-layer profiles (fused and per-layer DMFF); PARITY=1: the 16-bit parity table with the small-object mAP recipe (tools/parity16.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
if [ "$TESTS" = 1 ]; then      # the whole GPU suite + smoke on the library the evidence is taken with
  timeout 900 python -m pytest tests -q -m gpu --timeout=300 --tb=short -p no:cacheprovider > /tmp/ev_tests.log 2>&1
  echo "== GPU suite: $(tail -1 /tmp/ev_tests.log)"; grep -E "^(FAILED|ERROR)" /tmp/ev_tests.log | head -20
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
[ "$KEEP" = 1 ] || find gpurun_out -mindepth 1 -maxdepth 1 ! -name '.last_call.json' -exec rm -rf {} +     # the snapshot's old scratch: gpurun_out/ is capped at 64 MiB
WORKLOADS=${WORKLOADS:-"default c3 c4 c5"}
line () { python - "$1" "$2" <<'PY'
import json, sys
n, f = sys.argv[1:3]
try:
    d = json.load(open(f)); r = d["roofline"]
    print(n, {k: d.get(k) for k in ("value", "value_min", "value_max", "ms_per_step", "forward_ms_per_batch", "forward_only_pairs_per_s", "nms_ms_per_batch_standalone", "forward_roofline")},
          r["kernel"], r["frac"], r.get("avg_launch_us"), r["traffic"], r.get("algorithmic_bytes_per_launch"), (d.get("h2d_feed") or {}).get("pairs_per_s_with_h2d"))
except Exception as e:
    print(n, "FAILED", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
}
trace () {   # name, bench args ...: kernel stats of the bench command
  name=$1; shift
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/icaf_raw/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icaf_raw/prof_$name -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-h2d "$@" > $R/gpurun_out/prof_bench_$name.json 2> $R/gpurun_out/prof_$name.err
  f=$(find /tmp/icaf_raw/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_${name}_kernel_stats.csv && cut -c1-140 "$f" | sed -n 2,4p
  cd $R
}
sq () {      # suffix ("" = default workload), bench args ...: both SQ passes, merged and installed as profiles/pmc_sq<_suffix>.json for the bench line that follows
  suf=$1; shift
  cd $R && SQ_INSTS=1 PMC_NAME=$suf bash tools/gpu_pmc_sq.sh "$@" > gpurun_out/pmc_sq_${suf:-default}.log 2>&1
  grep "dmff\|cross_att\|stem\|sq summary" gpurun_out/pmc_sq_${suf:-default}.log | head -16
}
for w in $WORKLOADS; do
  case "$w" in
    default) name=""; ARGS="" ;;
    c3) name=c3_l_bf16_b32_640; ARGS="--model l --batch 32" ;;
    c4) name=c4_s_bf16_b64_512x640_loops3; ARGS="--loops 3 --height 512 --width 640 --batch 64" ;;
    c5) name=c5_l_vedai_f16_b16_1280; ARGS="--model l --dataset VEDAI --dtype f16 --height 1280 --width 1280 --batch 16 --conf 0.3" ;;
    *) echo "unknown workload $w"; continue ;;
  esac
  if [ -z "$name" ]; then
    cd $R && bash tools/gpu_pmc.sh 2>&1 | tail -2
    [ "${SQ:-1}" = 1 ] && sq ""
    cd $R && timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; line default gpurun_out/bench.json
    trace default
    trace depth1 --depth 1 --no-overlap
    cd $R && timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile.txt 2>/dev/null; head -1 gpurun_out/layer_profile.txt
    cd $R && ICAF_OPTIONS=dmff_fuse=0 timeout 300 python tools/layer_profile.py --autotune > gpurun_out/layer_profile_plain.txt 2>/dev/null; head -1 gpurun_out/layer_profile_plain.txt
    [ "$PARITY" = 1 ] && { timeout 1200 python tools/parity16.py --out gpurun_out/parity_16bit.json > gpurun_out/parity16.log 2>&1; tail -1 gpurun_out/parity16.log | cut -c1-200; }
    # the same workload on the fp32 build (the one held to north_star's 1e-3): what meeting that tolerance costs
    cd $R && timeout 900 python bench.py --dtype f32 --no-h2d --no-cpu-baseline --min-timed-seconds 1.5 > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err; line fp32 gpurun_out/bench_fp32.json
    # what a rank of an N > 1 run pays for the collective (one rank, RCCL group of one): plain / forced, twice, interleaved
    for r in 1 2; do for g in "" "--force-gather"; do
      timeout 300 python bench.py --no-cpu-baseline --no-latency --no-h2d --min-timed-seconds 1.5 $g 2> /dev/null | tail -1 > gpurun_out/bench_fg_${r}${g:+_gather}.json
      python -c "import json; d=json.load(open('gpurun_out/bench_fg_${r}${g:+_gather}.json')); print('force-gather' if '$g' else 'plain       ', d['value'], d['value_min'], d['value_max'])"
    done; done
    python - <<'PY'
import json
r = {k: [json.load(open(f"gpurun_out/bench_fg_{i}{s}.json"))["value"] for i in (1, 2)] for k, s in (("plain", ""), ("force_gather", "_gather"))}
r["ratio"] = round(sum(r["force_gather"]) / sum(r["plain"]), 4)
r["all_gather"] = json.load(open("gpurun_out/bench_fg_1_gather.json"))["config"]["all_gather"]
json.dump(r, open("gpurun_out/bench_force_gather.json", "w"), indent=1)
print("force-gather / plain =", r["ratio"])
PY
    # the driver's exact command, last
    cd $R && timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_form.json 2> gpurun_out/bench_driver_form.err; line driver_form gpurun_out/bench_driver_form.json
  else
    cp $R/profiles/tune_cache_$name.json /tmp/tune_$name.json       # (a re-tune inside the call edits the copy; it is brought back as gpurun_out/tune_cache_<name>.json)
    ARGS="$ARGS --tune-cache /tmp/tune_$name.json"
    cd $R && PMC_NAME=$name bash tools/gpu_pmc.sh $ARGS 2>&1 | tail -1
    [ "${SQ:-1}" = 1 ] && sq $name $ARGS
    cd $R && timeout 1200 python bench.py $ARGS > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; line $name gpurun_out/bench_$name.json
    trace $name $ARGS
    cd $R && timeout 300 python tools/layer_profile.py --autotune $ARGS > gpurun_out/layer_profile_$name.txt 2>/dev/null; head -1 gpurun_out/layer_profile_$name.txt
    cp /tmp/tune_$name.json gpurun_out/tune_cache_$name.json
  fi
done
