#!/usr/bin/env python3
"""Build a VARIANT of libicaf.so from a copy of icafusion_amd/csrc with some files replaced (kernel A/B on one GPU box:
run the same script twice with ICAF_LIB=<variant .so> / unset).

    python tools/build_variant.py <tag> [file=path_or_git_rev ...] [-DNAME=VALUE | -f<compiler flag> | <file>.hip:<flag for that file> ...]
e.g. python tools/build_variant.py oldepi conv_common.h=HEAD igemm.hip=HEAD        python tools/build_variant.py epifast -DICAF_EPI_FAST=1
writes icafusion_amd/lib/libicaf_<tag>.so (git-ignored, travels with the gpurun snapshot)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icafusion_amd import build as B   # noqa: E402

tag = sys.argv[1]
src = os.path.join(ROOT, f"_csrc_{tag}", "csrc")      # same depth as icafusion_amd/csrc (sources include ../../include/icaf.h)
shutil.rmtree(os.path.dirname(src), ignore_errors=True)
shutil.copytree(B.CSRC, src)
for spec in sys.argv[2:]:
    if ".hip:" in spec:                               # <file>.hip:<flag>: that file only
        f, flag = spec.split(":", 1)
        B.PER_FILE[f] = B.PER_FILE.get(f, []) + [flag]
        continue
    if spec.startswith("-"):
        B.COMMON = B.COMMON + [spec]                # every file of the variant is compiled with the define / compiler flag
        continue
    name, what = spec.split("=", 1)
    dst = os.path.join(src, name)
    if os.path.exists(what):
        shutil.copy(what, dst)
    else:
        open(dst, "w").write(subprocess.run(["git", "-C", ROOT, "show", f"{what}:icafusion_amd/csrc/{name}"],
                                            check=True, capture_output=True, text=True).stdout)
B.CSRC = src
B.OBJDIR = os.path.join(ROOT, "icafusion_amd", f"_obj_{tag}")
B.LIB = os.path.join(B.LIBDIR, f"libicaf_{tag}.so")
B.RESOURCES = os.path.join(B.LIBDIR, f"kernel_resources_{tag}.json")       # (not the product library's table)
B.build(force=True)
print(B.LIB)
