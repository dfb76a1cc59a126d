#!/bin/bash
# Round 4, third GPU call: the four-wavefront build of the three-launch block kernels (C = 128), the absolute oracle bound of the wide kernels,
# the small-object mAP recipe, the torchvision probe, the uint8 pipeline; P3 as two launches vs three (levels probe + whole forward, same box).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dmff_fused.py -q -m gpu --timeout=600 --tb=short -p no:cacheprovider -s > gpurun_out/t3a.log 2>&1; tail -1 gpurun_out/t3a.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t3a.log | sort | uniq -c | head -20
grep "vs float64 oracle" gpurun_out/t3a.log
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py -q -m gpu -k "uint8 or torchvision or pipeline" --timeout=600 --tb=short -p no:cacheprovider -s > gpurun_out/t3b.log 2>&1; tail -1 gpurun_out/t3b.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:|torchvision_present" gpurun_out/t3b.log | sort | uniq -c | head -20
timeout 900 python -m pytest tests/test_gpu_parity16.py -q -m gpu -k "ten_pixel or map50" --timeout=600 --tb=short -p no:cacheprovider -s > gpurun_out/t3c.log 2>&1; tail -1 gpurun_out/t3c.log
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/t3c.log | sort | uniq -c | head -20
python - <<'PY'
import json
for l in open("gpurun_out/t3c.log"):
    if l.startswith("{") and "map50_hip16" in l:
        d = json.loads(l); print(d["dtype"], d["recipe"][-12:], d.get("object_box_px"), "oracle", d["map50_oracle_fp32"], d["map_oracle_fp32"], "| ref16 d50", d["map50_delta_reference16"], "d", d["map_delta_reference16"], "| hip d50", d["map50_delta"], "d", d["map_delta"])
PY
echo "== levels"; timeout 300 python tools/probes/dmff_levels.py s 2>&1 | grep "^C=128"
echo "== forward A/B: P3 as two launches (default) vs three (ICAF_DMFF_FUSE_MAX_C=64)"
for mc in 128 64 128 64; do ICAF_DMFF_FUSE_MAX_C=$mc timeout 300 python tools/probes/ab_lib.py > gpurun_out/ab_mc$mc.json 2> gpurun_out/ab_mc$mc.err; python -c "
import json; d=json.load(open('gpurun_out/ab_mc$mc.json')); print('fuse_max_c=$mc forward_ms', [round(x,4) for x in d['forward_ms']], ' | '.join(f'{n.split()[0]}={t:.1f}' for n,t in d['launches'][22:27]))"; done
