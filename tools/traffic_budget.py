#!/usr/bin/env python3
"""HBM traffic budget of one forward: PMC bytes per dispatch (profiles/pmc_traffic*.json) x the launches of each kernel in the
per-launch profile (profiles/*layer_profile*.txt), next to the kernels' solo times.  CPU only; prints a table.

    python tools/traffic_budget.py [--pmc profiles/pmc_traffic.json] [--profile profiles/rNN_layer_profile_s_bf16_b32.txt]

Default profile: the NEWEST round's per-launch profile of the default workload (the PMC file and the profile must come from the same
kernels: a row whose kernel has no counters prints `?` and is left out of the sum, which the last line then says).

Kernel names group several layers (the PMC file holds the mean over the dispatches of a name), so a row is exact for the sum
over the layers of that name, not per layer."""
import argparse
import json
import os
import glob
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
_profiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_layer_profile_s_bf16_b32.txt")))
ap.add_argument("--profile", default=_profiles[-1] if _profiles else None)
a = ap.parse_args()
print(f"pmc: {os.path.relpath(a.pmc, ROOT)}   profile: {os.path.relpath(a.profile, ROOT)}")
pmc = json.load(open(a.pmc))["kernels"]
cnt, us = {}, {}
for line in open(a.profile).read().splitlines()[1:]:
    m = re.match(r"\s*\d+\s+([\d.]+) us\s+[\d.]+ TF\s+\d+ GB/s\s+(\S+)", line)
    if m:
        cnt[m.group(2)] = cnt.get(m.group(2), 0) + 1
        us[m.group(2)] = us.get(m.group(2), 0.0) + float(m.group(1))
rows, tot_b, tot_us = [], 0.0, 0.0
for name, n in cnt.items():
    v = pmc.get(name)
    b = (v["fetch_bytes_corrected"] + v["write_bytes_uncorrected"]) * n if v else None
    rows.append((name, n, us[name], b))
    if b:
        tot_b += b
        tot_us += us[name]
rows.sort(key=lambda r: -(r[3] or 0))
print(f"{'kernel':44s} launches  solo time   HBM bytes   rate")
for name, n, t, b in rows:
    print(f"{name:44s} x{n:2d}   {t:8.1f} us  " + (f"{b / 1e6:7.0f} MB  {b / t / 1e6:5.2f} TB/s" if b else "      ?"))
missing = [r[0] for r in rows if r[3] is None]
print(f"sum over kernels with counters: {tot_b / 1e9:.2f} GB in {tot_us:.0f} us = {tot_b / tot_us / 1e6:.2f} TB/s "
      f"(all launches: {sum(us.values()):.0f} us" + (f"; NO counters for {len(missing)} kernel names: {', '.join(missing)}" if missing else "") + ")")
