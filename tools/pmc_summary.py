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


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"icaf::(\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:80]


def collect(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", "?"))
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0))
                e = out.setdefault(k, {}).setdefault(c, [0.0, 0])
                e[0] += v
                e[1] += 1
    return out


def main():
    res = {}
    for d in sys.argv[1:]:
        for k, cs in collect(d).items():
            for c, (s, n) in cs.items():
                res.setdefault(k, {})[c] = {"mean_per_dispatch": s / max(n, 1), "dispatches": n}
    for k, cs in res.items():
        if "FETCH_SIZE" in cs:   # KB units (rocprofv3 derived metric: 64 B requests / 1024)
            cs["fetch_bytes_corrected"] = cs["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
        if "WRITE_SIZE" in cs:
            cs["write_bytes_uncorrected"] = cs["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
