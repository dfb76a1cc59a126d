#!/bin/bash
# Round 4, second GPU call: hidden-split proj_mlp at P5 (tests + level timings + forward A/B by ICAF_DMFF_KSPLIT), the uint8 host feed
# through staging buffers, and the c3 (yolov5l) level timings.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py tests/test_gpu_pipeline.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s > gpurun_out/t2.log 2>&1; tail -1 gpurun_out/t2.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t2.log | sort | uniq -c | head -20; grep "ksplit=" gpurun_out/t2.log | head -12
for ks in 1 2 4; do echo "== levels ICAF_DMFF_KSPLIT=$ks"; ICAF_DMFF_KSPLIT=$ks timeout 300 python tools/probes/dmff_levels.py s 2>&1 | grep "three"; done
echo "== levels l, auto"; timeout 300 python tools/probes/dmff_levels.py l 2>&1 | grep "three"
for ks in 2 4; do echo "== levels l ICAF_DMFF_KSPLIT=$ks"; ICAF_DMFF_KSPLIT=$ks timeout 300 python tools/probes/dmff_levels.py l 2>&1 | grep "three"; done
echo "== forward A/B"
for ks in 1 0; do ICAF_DMFF_KSPLIT=$ks timeout 300 python tools/probes/ab_lib.py > gpurun_out/ab_ks$ks.json 2> gpurun_out/ab_ks$ks.err; done
python - <<'PY'
import json
a, b = json.load(open("gpurun_out/ab_ks1.json")), json.load(open("gpurun_out/ab_ks0.json"))
print("forward_ms unsplit:", a["forward_ms"], " auto:", b["forward_ms"])
for n, t in b["launches"]:
    if "dmff" in n or "attention" in n: print(f"   {n[:60]:60s} {t:7.1f} us")
PY
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline --no-latency > gpurun_out/bench2.json 2> gpurun_out/bench2.err; tail -3 gpurun_out/bench2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench2.json"))
print("value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
print("h2d", d.get("h2d_feed"))
PY
