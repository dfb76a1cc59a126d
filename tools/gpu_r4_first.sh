#!/bin/bash
# Round 4, first GPU call: the whole GPU suite on the new library (attention loop of attn_core.h, MFMA results in VGPRs for the DMFF files,
# per-device LDS opt-in in every launcher, rank-major pipeline outputs), then same-box A/Bs:
#   r3dmff  = this tree with round 3's dmff.hip / dmff_fused.hip            (what the attention rewrite bought)
#   vform   = this tree, EVERY file compiled with -amdgpu-mfma-vgpr-form     (is that form a win for the convolution kernels too?)
# then the bench line with its new extras (whole-forward PMC traffic, overlapped launch time, PCIe-inclusive feed), and the SQ counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1; tail -1 gpurun_out/env.log
python -c "import torchvision; print('torchvision', torchvision.__version__)" > gpurun_out/torchvision.log 2>&1; tail -1 gpurun_out/torchvision.log
bash tools/gpu_tests.sh 2>&1 | tail -40
echo "== DMFF levels (new library)"; timeout 300 python tools/probes/dmff_levels.py s 2>&1 | grep "^C=" ; timeout 300 python tools/probes/dmff_levels.py l 2>&1 | grep "^C="
for v in r3dmff vform; do
  if [ -f icafusion_amd/lib/libicaf_$v.so ]; then
    echo "== DMFF levels ($v)"; ICAF_LIB=$R/icafusion_amd/lib/libicaf_$v.so timeout 300 python tools/probes/dmff_levels.py s 2>&1 | grep "^C="
  fi
done
echo "== forward A/B"
timeout 300 python tools/probes/ab_lib.py > gpurun_out/ab_new.json 2> gpurun_out/ab_new.err
for v in r3dmff vform; do
  [ -f icafusion_amd/lib/libicaf_$v.so ] && ICAF_LIB=$R/icafusion_amd/lib/libicaf_$v.so timeout 300 python tools/probes/ab_lib.py > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
done
python - <<'PY'
import json
def load(t):
    try: return json.load(open(f"gpurun_out/ab_{t}.json"))
    except Exception as e: return None
new = load("new")
print("forward_ms new:", new and new["forward_ms"])
for t in ("r3dmff", "vform"):
    o = load(t)
    if not (o and new): continue
    print(f"forward_ms {t}:", o["forward_ms"])
    for (n, a), (_, b) in zip(new["launches"], o["launches"]):
        if abs(a - b) > 0.03 * max(a, b) and abs(a - b) > 1.0:
            print(f"   {n[:70]:70s} new {a:7.1f} us   {t} {b:7.1f} us")
PY
echo "== bench"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json")); r = d["roofline"]
print("value", d["value"], d["value_min"], d["value_max"], "fwd_ms", d["forward_ms_per_batch"], "fwd-only", d["forward_only_pairs_per_s"], d["forward_only_pairs_per_s_one_in_flight"])
print("roofline", r["kernel"], r["frac"], r["avg_launch_us"], r.get("avg_launch_us_with_second_forward_in_flight"), r["traffic"])
print("forward_roofline", d["forward_roofline"])
print("h2d", d.get("h2d_feed"))
print("latency", d.get("latency_b1"))
for k, v in list(d["kernels"].items())[:40]: print(f"  {k:44s} {v}")
PY
echo "== SQ counters"
SQ_INSTS=1 bash tools/gpu_pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1; grep -c mfma_util gpurun_out/pmc_sq.log; grep "dmff\|cross_att" gpurun_out/pmc_sq.log
