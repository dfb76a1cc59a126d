#!/usr/bin/env python3
"""Static instruction mix of the kernels a workload launches — the analysis behind docs/HISTORY.md section 15's "bound by vector-instruction issue":

    python tools/isa_mix.py [profiles/r04_bench_s_bf16_b32_depth1_kernel_stats.csv] > profiles/r04_isa_mix_s_bf16_b32.txt

Compiles every csrc/*.hip to gfx950 assembly with the library's own flags (device code only, no GPU needed), splits it by kernel, and counts per
kernel the MFMA / transcendental / other vector / LDS / vector-memory / scalar instructions, `s_waitcnt` and `s_barrier` — STATIC counts (a loop body
counts once), next to the registers / occupancy of the build (icafusion_amd/lib/kernel_resources.json) and, if a rocprofv3 kernel-stats csv is given,
only for the kernels that file names, with their call counts and average durations.  Read it with the SQ counters of profiles/r04_pmc_sq_insts.json
(dynamic instruction counts, VALU issue fraction): a kernel with a high VALU issue fraction whose vector instructions are mostly NOT its arithmetic
(SiLU = 5 per value: mul, v_exp, add, v_rcp, mul; a bf16 pair conversion; an MFMA) is spending them on address arithmetic — stem2 and icaf_bottleneck were."""
import collections
import concurrent.futures as cf
import csv
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icafusion_amd import build as B   # noqa: E402


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if re.match(r"v_(exp|rcp|rsq|sqrt|log|sin|cos)_", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op == "s_waitcnt":
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def asm_of(src, outdir):
    out = os.path.join(outdir, src.replace(".hip", ".s"))
    flags = [c for c in B.COMMON if c not in ("-fPIC", "-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage")] + B.PER_FILE.get(src, [])
    subprocess.run([B.hipcc()] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, src)], check=True, capture_output=True)
    return out


def kernels_of(path):
    """mangled kernel name -> Counter of instruction classes (amdhsa kernels only)"""
    text = open(path).read()
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M))
    res, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(\S+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur = m.group(1) if m.group(1) in names else None
            if cur:
                res[cur] = collections.Counter()
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):                 # (not the first s_endpgm: a kernel with a uniform early exit has several)
            cur = None
        if cur is None or not t or t[0] in ";." or t.endswith(":"):
            continue
        res[cur][classify(t.split()[0])] += 1
    return res


def main():
    wanted = None
    if len(sys.argv) > 1:
        wanted = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(sys.argv[1])) if "icaf::" in r["Name"]}
    with tempfile.TemporaryDirectory() as td, cf.ThreadPoolExecutor(8) as ex:
        counts = {}
        for path in ex.map(lambda s: asm_of(s, td), B.sources()):
            counts.update(kernels_of(path))
    pretty = B.demangle(sorted(counts))
    resources = json.load(open(B.RESOURCES)) if os.path.exists(B.RESOURCES) else {}
    rows = []
    for k, c in counts.items():
        name = pretty[k]
        if wanted is not None and name not in wanted:
            continue
        r = resources.get(name[:400], {})
        rows.append((wanted[name] if wanted else (0, 0.0), name, c, r))
    rows.sort(key=lambda x: -(x[0][0] * x[0][1]))
    cols = ("mfma", "valu", "trans", "lds", "vmem", "salu", "waitcnt", "barrier")
    print(f"{'calls':>6} {'avg us':>8} " + " ".join(f"{c:>7}" for c in cols) + "  valu+trans/mfma  vgpr agpr occ  kernel")
    for (calls, us), name, c, r in rows:
        per = (c["valu"] + c["trans"]) / c["mfma"] if c["mfma"] else float("nan")
        print(f"{calls:6d} {us:8.1f} " + " ".join(f"{c[x]:7d}" for x in cols) + f"  {per:15.1f}  {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('occupancy', 0):3d}  "
              + name.replace("icaf::", "")[:150])


if __name__ == "__main__":
    main()
