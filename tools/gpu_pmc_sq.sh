#!/bin/bash
# MFMA-pipe utilisation per kernel from the SQ counters (one pass): SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed
# over all SIMDs, SQ_BUSY_CYCLES the kernel's busy cycles summed over the 32 shader engines (MI355X_MICROARCH.md).
#   util = MFMA_BUSY / (SQ_BUSY / 32 * 1024 SIMDs)
# PMC_NAME=<suffix>: the merged summary of both passes (with the bench line's `workload` string) is written to gpurun_out/pmc_sq_merged<_suffix>.json AND installed
# as profiles/pmc_sq<_suffix>.json in the box's copy, so that a bench.py run later in the same call carries roofline.valu_issue_frac / mfma_busy / wave_wait_frac.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
SUF=${PMC_NAME:+_$PMC_NAME}
cd /tmp && export TMPDIR=/tmp
RAW=/tmp/icaf_raw; mkdir -p $RAW
rm -rf $RAW/pmc_sq; rm -f $R/gpurun_out/pmc_sq_summary.json $R/gpurun_out/pmc_sq_insts.json
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $RAW/pmc_sq -o pmc -- \
    python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --no-graph --no-overlap --depth 1 --steps 3 --warmup 1 --repeats 1 "$@" > $R/gpurun_out/pmc_sq.json 2> $R/gpurun_out/pmc_sq.err
tail -2 $R/gpurun_out/pmc_sq.err
cd $R && python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from pmc_summary import short
f = glob.glob("/tmp/icaf_raw/pmc_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
out = {}
for k, c in acc.items():
    if k.startswith("at::") or k.startswith("__") or "rocprim" in k or not c["SQ_BUSY_CYCLES"]: continue
    out[k] = {"dispatches": n[k], "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] / 32 * 1024), 4),
              "wave_wait_frac": round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3),
              "wave_issue_stall_frac": round(c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3),
              "wave_active_frac": round(c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3)}
json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; mfma_util = MFMA_BUSY / (SQ_BUSY / 32 * 1024)", "kernels": out}, open("gpurun_out/pmc_sq_summary.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_util"]): print(f"{k:42s} {v}")
PY
# Second pass (SQ_INSTS=1): instruction mix per kernel — what bounds the MFMA-busy fraction from above.  With VALU arithmetic and MFMAs
# of one SIMD perfectly overlapped, the pipe can be busy at most MFMA_cycles / max(MFMA_cycles, VALU_issue_cycles) of the time:
#   mfma_cycles_per_simd = MFMA_BUSY / 1024;  valu_issue_cycles_per_simd = 4 * SQ_ACTIVE_INST_VALU / 1024  (quad-cycles -> cycles)
if [ "${SQ_INSTS:-0}" = "1" ]; then
cd /tmp
rm -rf $RAW/pmc_sq2
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $RAW/pmc_sq2 -o pmc -- \
    python $R/bench.py --no-cpu-baseline --no-latency --no-h2d --no-graph --no-overlap --depth 1 --steps 3 --warmup 1 --repeats 1 "$@" > $R/gpurun_out/pmc_sq2.json 2> $R/gpurun_out/pmc_sq2.err
tail -2 $R/gpurun_out/pmc_sq2.err
cd $R && python - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from pmc_summary import short
f = glob.glob("/tmp/icaf_raw/pmc_sq2/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
out = {}
for k, c in acc.items():
    if k.startswith("at::") or k.startswith("__") or "rocprim" in k or not c["SQ_BUSY_CYCLES"]: continue
    simd_cycles = c["SQ_BUSY_CYCLES"] / 32 * 1024
    mf, va = c["SQ_VALU_MFMA_BUSY_CYCLES"], 4.0 * c["SQ_ACTIVE_INST_VALU"]
    out[k] = {"dispatches": n[k], "insts_valu": c["SQ_INSTS_VALU"] / n[k], "insts_mfma": c["SQ_INSTS_MFMA"] / n[k], "insts_trans_f32": c["SQ_INSTS_VALU_TRANS_F32"] / n[k],
              "insts_lds": c["SQ_INSTS_LDS"] / n[k], "insts_salu": c["SQ_INSTS_SALU"] / n[k],
              "valu_per_mfma": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2) if c["SQ_INSTS_MFMA"] else None,
              "mfma_util": round(mf / simd_cycles, 4), "valu_issue_frac": round(va / simd_cycles, 4),
              "mfma_util_ceiling_if_valu_fully_overlapped": round(mf / max(mf, va), 4) if mf else 0.0}
json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; "
                     "per dispatch means; valu_issue_frac = 4 * ACTIVE_INST_VALU / (SQ_BUSY / 32 * 1024); ceiling = MFMA_BUSY / max(MFMA_BUSY, 4 * ACTIVE_INST_VALU)", "kernels": out},
          open("gpurun_out/pmc_sq_insts.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_util"]): print(f"{k:42s} valu/mfma {v['valu_per_mfma']} mfma {v['mfma_util']} valu_issue {v['valu_issue_frac']} ceiling {v['mfma_util_ceiling_if_valu_fully_overlapped']}")
PY
fi
cd $R && python - "$SUF" <<'PY'
import json, os, sys
suf = sys.argv[1]
a = json.load(open("gpurun_out/pmc_sq_summary.json"))
b = json.load(open("gpurun_out/pmc_sq_insts.json")) if os.path.exists("gpurun_out/pmc_sq_insts.json") else {"kernels": {}, "method": ""}
try:
    w = json.load(open("gpurun_out/pmc_sq.json"))["config"]["workload"]
except Exception:
    w = None
k = {}
for name in sorted(set(a["kernels"]) | set(b["kernels"])):
    k[name] = dict(b["kernels"].get(name, {}), **a["kernels"].get(name, {}))
out = {"workload": w, "method": [a["method"], b["method"]], "kernels": k}
json.dump(out, open(f"gpurun_out/pmc_sq_merged{suf}.json", "w"), indent=1, sort_keys=True)
json.dump(out, open(f"profiles/pmc_sq{suf}.json", "w"), indent=1, sort_keys=True)
print("sq summary for:", w)
PY
