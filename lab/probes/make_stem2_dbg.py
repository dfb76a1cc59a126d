#!/usr/bin/env python3
"""Build libicaf_dbg.so: the current sources with stage switches compiled into icaf_stem2 (env ICAF_STEM2_DBG, bits:
1 no prefetch / commit, 2 no stage 1, 4 no stage 2, 8 no stage 3, 16 no global stores — results are then garbage, the time
of the remaining stages is what is measured):

    python lab/probes/make_stem2_dbg.py
    gpurun -- 'export ICAF_LIB=$PWD/icafusion_amd/lib/libicaf_dbg.so; for d in 0 1 2 4 8 16 30 31; do ICAF_STEM2_DBG=$d python lab/probes/stem2_ablation.py; done'
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "icafusion_amd", "csrc", "stem.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old[:70]
    s = s.replace(old, new)


rep("    int Hs, Ws;                 // the stem's output size (H/2, W/2)\n", "    int Hs, Ws;\n    int dbg;\n")
rep("        if (more) fetch(nxt_t, v0, v1);            // next tile's image reads stay in flight during everything below\n",
    "        if (more && !(q.dbg & 1)) fetch(nxt_t, v0, v1);\n")
rep("        {\n            for (int j = wave; j < (S2_NH + 31) / 32; j += S2_THREADS / 64) {",
    "        if (!(q.dbg & 2)) {\n            for (int j = wave; j < (S2_NH + 31) / 32; j += S2_THREADS / 64) {")
rep("#pragma unroll\n        for (int k = 0; k < 9 * C0 / 16; ++k) {", "if (!(q.dbg & 4))\n#pragma unroll\n        for (int k = 0; k < 9 * C0 / 16; ++k) {")
rep("#pragma unroll\n        for (int s2 = 0; s2 < C1 / 16; ++s2) {", "if (!(q.dbg & 8))\n#pragma unroll\n        for (int s2 = 0; s2 < C1 / 16; ++s2) {")
rep("            if (more) commit(v0, v1);", "            if (more && !(q.dbg & 1)) commit(v0, v1);")
rep("                if (yoff[it] >= 0) *(u32x4*)(yg + yoff[it]) = sv[it];", "                if (yoff[it] >= 0 && !(q.dbg & 16)) *(u32x4*)(yg + yoff[it]) = sv[it];")
rep("    q.Hs = a->H / 2; q.Ws = a->W / 2;\n", "    q.Hs = a->H / 2; q.Ws = a->W / 2;\n    q.dbg = getenv(\"ICAF_STEM2_DBG\") ? atoi(getenv(\"ICAF_STEM2_DBG\")) : 0;\n")
s = s.replace("#include <cstring>\n", "#include <cstring>\n#include <cstdlib>\n", 1)
with tempfile.NamedTemporaryFile("w", suffix=".hip", delete=False) as f:
    f.write(s)
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "dbg", f"stem.hip={f.name}"]))
