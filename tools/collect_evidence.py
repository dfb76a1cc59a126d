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
         ("pmc_sq_summary_default.json", f"{tag}_pmc_sq_mfma_util.json"), ("pmc_sq_insts_default.json", f"{tag}_pmc_sq_insts.json"),
         ("pmc_sq_summary_c3.json", f"{tag}_pmc_sq_mfma_util_{CFG['c3']}.json"), ("pmc_sq_insts_c3.json", f"{tag}_pmc_sq_insts_{CFG['c3']}.json"),
         ("pmc_sq_summary_c3_pers.json", f"{tag}_pmc_sq_mfma_util_{CFG['c3']}_with_igemm_pers.json"),
         ("pmc_sq_insts_c3_pers.json", f"{tag}_pmc_sq_insts_{CFG['c3']}_with_igemm_pers.json"),
         ("pmc_summary.json", "pmc_traffic.json"), ("parity_16bit.json", "parity_16bit.json")]
for k, n in CFG.items():
    pairs += [(f"bench_{n}.json", f"{tag}_bench_{n}.json"), (f"prof_{n}_kernel_stats.csv", f"{tag}_bench_{n}_kernel_stats.csv"),
              (f"pmc_summary_{n}.json", f"pmc_traffic_{n}.json")]
done = []
for src, dst in pairs:
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, dst))
        done.append(dst)
print("copied:", *done, sep="\n  ")
