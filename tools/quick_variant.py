#!/usr/bin/env python3
"""Fast variant of libicaf.so for same-box kernel A/B: recompile ONLY the named sources (with extra -D defines) and link them with the
objects of the last full build (icafusion_amd/_obj).  tools/build_variant.py rebuilds everything (3 min); this takes one compile.

    python tools/quick_variant.py <tag> <file.hip>[,<file.hip>...] [-DNAME=VALUE ...]
writes icafusion_amd/lib/libicaf_<tag>.so (git-ignored; travels with the gpurun snapshot; select it with ICAF_LIB)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from icafusion_amd import build as B   # noqa: E402

tag, files, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
objdir = os.path.join(ROOT, "icafusion_amd", f"_obj_{tag}")
os.makedirs(objdir, exist_ok=True)
objs = []
for src in B.sources():
    base = os.path.join(B.OBJDIR, src.replace(".hip", ".o"))
    if src in files:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [B.hipcc()] + [c for c in B.COMMON if c not in ("-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage")] + B.PER_FILE.get(src, []) + defs + \
              ["-c", os.path.join(B.CSRC, src), "-o", obj]
        subprocess.run(cmd, check=True)
        objs.append(obj)
    else:
        assert os.path.exists(base), f"{base}: run python -m icafusion_amd.build first"
        objs.append(base)
lib = os.path.join(B.LIBDIR, f"libicaf_{tag}.so")
subprocess.run([B.hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
