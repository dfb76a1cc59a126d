#!/usr/bin/env python3
"""Copy the summaries a tools/gpu_evidence.sh call left in gpurun_out/ into profiles/ under the round's prefix (the files the judge reads are the
tracked ones under profiles/; gpurun_out/ is scratch).      python tools/collect_evidence.py r05"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
CFG = {"c3": "c3_l_bf16_b32_640", "c4": "c4_s_bf16_b64_512x640_loops3", "c5": "c5_l_vedai_f16_b16_1280"}
pairs = [("bench.json", f"{tag}_bench_s_bf16_b32.json"), ("prof_default_kernel_stats.csv", f"{tag}_bench_s_bf16_b32_kernel_stats.csv"),
         ("prof_depth1_kernel_stats.csv", f"{tag}_bench_s_bf16_b32_depth1_kernel_stats.csv"),
         ("layer_profile.txt", f"{tag}_layer_profile_s_bf16_b32.txt"), ("layer_profile_plain.txt", f"{tag}_layer_profile_s_bf16_b32_dmff_per_layer.txt"),
         # the SQ passes (merged: MFMA busy, VALU issue, wave-parked fractions, instruction mix): the CURRENT summary bench.py reads + the round's copy
         ("pmc_sq_merged.json", "pmc_sq.json"), ("pmc_sq_merged.json", f"{tag}_pmc_sq.json"),
         ("pmc_summary.json", "pmc_traffic.json"), ("parity_16bit.json", "parity_16bit.json"),
         ("bench_fp32.json", f"{tag}_bench_s_fp32_b32.json"), ("bench_driver_form.json", f"{tag}_bench_driver_form.json"),
         ("bench_force_gather.json", f"{tag}_bench_force_gather.json")]
for k, n in CFG.items():
    pairs += [(f"bench_{n}.json", f"{tag}_bench_{n}.json"), (f"prof_{n}_kernel_stats.csv", f"{tag}_bench_{n}_kernel_stats.csv"),
              (f"pmc_summary_{n}.json", f"pmc_traffic_{n}.json"), (f"pmc_sq_merged_{n}.json", f"pmc_sq_{n}.json"), (f"pmc_sq_merged_{n}.json", f"{tag}_pmc_sq_{n}.json"),
              (f"layer_profile_{n}.txt", f"{tag}_layer_profile_{n}.txt")]
done = []
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
        done.append(dst)
print("copied:", *done, sep="\n  ")
