#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel mean bytes per dispatch.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the TCC EA request counters; following
MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 tallies 128-byte requests at 64 B, so wide coalesced reads are
doubled here (`fetch_bytes_corrected`); WRITE_SIZE is reported uncorrected (uncalibrated per that guide)."""
import csv
import glob
import json
import os
import re
import sys


_DN = ["f32", "bf16", "f16"]


def short(name):
    """rocprof kernel name -> the name bench.py reports (igemm instantiations that differ only in the activation
    template argument are merged, as ops.conv_kernel_name does)."""
    name = re.sub(r"^void ", "", name)
    m = re.match(r"icaf::igemm_dma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), \d+, (\d+), (\d+)(?:, \w+)*>", name)
    if m:
        dt, odt, bm, bn, wm, wn, rb, ns = map(int, m.groups())
        w8 = "w8" if bm == 128 and (bm // wm) * (bn // wn) == 8 else ""          # 8-wavefront build of a 128-row tile
        return f"igemm_dma{rb}x{ns}_{_DN[dt]}_{_DN[odt]}_{bm}x{bn}{w8}"
    m = re.match(r"icaf::igemm_stream_kernel<(\d+), (\d+)(?:, \w+)*>", name)         # persistent streaming GEMM (DT, BN, ACT, MODE, PRE)
    if m:
        dt, bn = map(int, m.groups())
        return f"igemm_stream_{_DN[dt]}_128x{bn}"
    m = re.match(r"icaf::igemm_wreg_kernel<(\d+), (\d+), \d+, \d+(?:, (\d+))?(?:, (\d+))?>", name)           # weight operand fed from registers (DT, NWV, ACT, MODE, TN, BM)
    if m:
        dt, nwv = int(m.group(1)), int(m.group(2))
        tn, bm = int(m.group(3) or 1), int(m.group(4) or 128)
        tag = {(4, 1, 128): "128x128", (8, 1, 128): "128x256", (8, 2, 128): "128x512", (4, 2, 128): "128x256w4", (4, 2, 64): "64x256", (4, 1, 64): "64x128"}.get((nwv, tn, bm), f"{bm}x{32 * nwv * tn}")
        return f"igemm_wreg_{_DN[dt]}_{tag}"                                    # (= wreg_tag() of igemm_wreg.hip, what bench.py reports)
    m = re.match(r"icaf::cstream_kernel<(\d+), (true|false)>", name)              # persistent 3x3, filter resident in LDS
    if m:
        return f"cstream_{_DN[int(m.group(1))]}_8x16n64"                        # (with or without the chained 1x1: one name, as bench.py reports it)
    m = re.match(r"icaf::cwide_kernel<(\d+), (\d+), (\d+), (\d+)(?:, \w+)*>", name)      # 3x3 from a resident halo patch, weights into registers
    if m:
        dt, cin, st, nsub = map(int, m.groups())
        return f"cwide_{_DN[dt]}_8x{16 if nsub == 4 else 8}n128" + ("s2" if st == 2 else "") + ("c64" if cin == 64 else "")
    if name.startswith("icaf::detect_conv_kernel<"):
        return "detect_conv+decode"
    m = re.match(r"icaf::igemm_kernel<(\d+), (\d+), (\d+), (\d+),", name)
    if m:
        dt, odt, bm, bn = map(int, m.groups())
        return f"igemm_reg_{_DN[dt]}_{_DN[odt]}_{bm}x{bn}"
    m = re.match(r"icaf::ctile_kernel<(\d+), (\d+), (\d+), (\d+), \d+, (\d+)>", name)
    if m:
        dt, th, tw, bn, st = map(int, m.groups())
        return f"ctile_{_DN[dt]}_{th}x{tw}n{bn}" + ("s2" if st == 2 else "")
    if re.match(r"icaf::bneck_kernel<[^>]*, true>", name):
        return "bottleneck+cv3"
    m = re.match(r"icaf::(\w+)(<[^>]*>)?", name)
    if m:
        return {"preprocess_kernel": "preprocess_s2d", "pool_tokens_kernel": "dmff_pool_tokens", "upsample_merge_kernel": "dmff_upsample_merge",
                "cross_attn_kernel": "cross_attention", "detect_decode_kernel": "detect_decode", "detect_pixel_kernel": "detect_decode", "sppf_lds_kernel": "sppf_pool",
                "upsample_kernel": "upsample_nearest", "stem_kernel": "stem", "bneck_kernel": "bottleneck",
                "stem2_kernel": "stem+conv3x3s2+1x1", "pool_tokens_rows_kernel": "dmff_pool_tokens",
                "dmff_attn_mlp_kernel": "dmff_attn_mlp", "dmff_ln_qkv_kernel": "dmff_ln_qkv", "layernorm_kernel": "layernorm",
                "dmff_wide_ln_qkv_kernel": "dmff_ln_qkv", "dmff_wide_proj_mlp_kernel": "dmff_proj_mlp",
                "dmff_wide_reduce_kernel": "dmff_proj_mlp_reduce"}.get(m.group(1), m.group(1))
    return name[:80]


def collect(d, by_instantiation=False):
    """by_instantiation: key the DMFF block kernels by their FULL template name instead (the three levels of a model run different
    instantiations of one kernel — C = 128 four wavefronts, C = 256, C = 512 with the hidden split — which the short name merges)."""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                full = row.get("Kernel_Name", "?")
                if by_instantiation:
                    if "dmff_wide" not in full and "dmff_attn_mlp" not in full and "dmff_ln_qkv" not in full:
                        continue
                    k = re.sub(r"^void ", "", full).replace("icaf::", "")[:120]
                else:
                    k = short(full)
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0))
                e = out.setdefault(k, {}).setdefault(c, [0.0, 0])
                e[0] += v
                e[1] += 1
    return out


def main():
    res = {}
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--workload=")), None)
    for d in dirs:
        for k, cs in collect(d).items():       # collect() already merges kernels that map to the same short name
            for c, (s, n) in cs.items():
                res.setdefault(k, {})[c] = {"mean_per_dispatch": s / max(n, 1), "dispatches": n}
    for k, cs in res.items():
        if "FETCH_SIZE" in cs:   # KB units (rocprofv3 derived metric: 64 B requests / 1024)
            cs["fetch_bytes_corrected"] = cs["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
        if "WRITE_SIZE" in cs:
            cs["write_bytes_uncorrected"] = cs["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    res = {k: v for k, v in res.items() if not k.startswith("at::")}
    inst = {}
    for d in dirs:
        for k, cs in collect(d, by_instantiation=True).items():
            for c, (s, n) in cs.items():
                inst.setdefault(k, {})[c] = {"mean_per_dispatch": s / max(n, 1), "dispatches": n}
    for k, cs in inst.items():
        if "FETCH_SIZE" in cs:
            cs["fetch_bytes_corrected"] = cs["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
        if "WRITE_SIZE" in cs:
            cs["write_bytes_uncorrected"] = cs["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    print(json.dumps({"workload": workload, "unit": "bytes per dispatch", "dmff_kernels_by_instantiation": inst,
                      "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                                "FETCH_SIZE x1024 x2 (gfx950 128-byte requests tallied at 64 B), WRITE_SIZE x1024",
                      "kernels": res}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
