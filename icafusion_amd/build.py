"""Build libicaf.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot to the GPU box).

    python -m icafusion_amd.build [--force]

One object per .hip file (parallel), then one shared library exporting the C ABI of include/icaf.h.
detect.hip / nms.hip are built with fp contraction off: their arithmetic must round like the reference's
separate fp32 torch ops so that NMS keep-indices are bit-exact.

Every compile also asks the backend for its per-kernel resource usage (-Rpass-analysis=kernel-resource-usage); the
summary lands in lib/kernel_resources.json, and the build FAILS if a kernel that synchronises its LDS-DMA ring with
counted `s_waitcnt vmcnt(N)` waits (igemm_dma / ctile / bneck / stem / stem2 / dmff_* kernels) uses scratch memory: a register
spill is a VMEM operation too, it bumps the same counter, and the counted wait would then let a wave read a slice of
the ring that has not landed yet — silently, and only at large grids (docs/HISTORY.md §10).
"""
import concurrent.futures as cf
import hashlib
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "_obj")
LIB = os.path.join(LIBDIR, "libicaf.so")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-comment",
          "-Rpass-analysis=kernel-resource-usage"]
RESOURCES = os.path.join(LIBDIR, "kernel_resources.json")
# kernels whose K loops use counted vmcnt waits: any scratch use (spill) would race with them
NO_SCRATCH = re.compile(r"cstream_kernel|cwide_kernel|detect_conv_kernel|igemm_dma_kernel|igemm_stream_kernel|igemm_wreg_kernel|ctile_kernel|bneck_kernel|stem_kernel|stem2_kernel|dmff_\w*kernel")
_REMARK = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                     r"LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill):\s+(\S+)")


def parse_resources(stderr):
    """-Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgpr, agpr, sgpr, scratch, occupancy, lds, ...}}"""
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "LDS Size [bytes/block]": "lds", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}
    for k, v in _REMARK.findall(stderr):
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[keys[k]] = int(v)
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs instead of AGPRs (gfx950: one unified register file).  Kernels whose MFMA results
# feed VALU arithmetic directly (softmax, GELU) otherwise pay one v_accvgpr_read per accumulator element and tile — the attention loop
# spent 32 of ~118 issue slots per score tile on them (attn_core.h) — and AGPR-allocated accumulators round the register budget up.
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# (ctile.hip, round 6: the fused Bottleneck + cv3 kernel issues 112 v_accvgpr_read per wave for its seven accumulators and sits at VALU issue 0.70:
#  115 registers and no AGPRs instead of 100 + 64, 135 -> 133 us; the other convolution files already compile to VGPR accumulators or do not move)
PER_FILE = {"detect.hip": ["-ffp-contract=off"], "nms.hip": ["-ffp-contract=off"],
            "dmff.hip": VGPR_FORM, "dmff_fused.hip": VGPR_FORM, "ctile.hip": VGPR_FORM}
_VF = os.environ.get("ICAF_VGPR_FORM", "")             # A/B builds (tools/build_variant.py): "all" = the whole library in that form, "none" = no file
if _VF == "all":                                       # (the file list IS the source directory: a new .hip file is never left out)
    PER_FILE = {f: [x for x in PER_FILE.get(f, []) if x not in VGPR_FORM] + VGPR_FORM
                for f in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))}
elif _VF == "none":
    PER_FILE = {f: [x for x in v if x not in VGPR_FORM] for f, v in PER_FILE.items()}
EXPORT = "-fvisibility=default"


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


_CC_ID = {}


def compiler_id(cc):
    """`hipcc --version` (compiler + ROCm release): part of every object's cache key, so that a compiler upgrade — or another HIPCC — never reuses
    stale objects and their cached register / scratch report (which gates the NO_SCRATCH check)."""
    if cc not in _CC_ID:
        try:
            _CC_ID[cc] = subprocess.run([cc, "--version"], capture_output=True, text=True, check=True).stdout.strip()
        except Exception as e:
            _CC_ID[cc] = f"{cc}: {e}"
    return _CC_ID[cc]


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "icaf.h"))
    srcs = sources()
    stamp = LIB[:-3] + ".stamp"
    want = _digest([os.path.join(CSRC, s) for s in srcs] + headers, " ".join(COMMON) + repr(sorted(PER_FILE.items())) + compiler_id(hipcc()))
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        if verbose:
            print(f"[icafusion_amd.build] {LIB} is up to date")
        return LIB
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        # visibility: the extern "C" entry points carry default visibility through the flag below
        cmd = [cc] + [c for c in COMMON if c != "-fvisibility=hidden"] + PER_FILE.get(src, []) + \
              ["-c", os.path.join(CSRC, src), "-o", obj]
        # per-object cache: an object is reused when its source, every header and its flags are unchanged (a kernel iteration recompiles ONE
        # file instead of fifteen: 20-200 s instead of 3.5 min); the resource remarks of the compile are kept beside it
        key = _digest([os.path.join(CSRC, src)] + headers, " ".join(cmd[1:]) + "\n" + compiler_id(cc))
        keyf, resf = obj[:-2] + ".key", obj[:-2] + ".res.json"
        if not force and os.path.exists(obj) and os.path.exists(keyf) and os.path.exists(resf) and open(keyf).read().strip() == key:
            return obj, json.load(open(resf))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr[-6000:]}")
        res1 = parse_resources(r.stderr)
        with open(resf, "w") as f:
            json.dump(res1, f)
        with open(keyf, "w") as f:
            f.write(key)
        return obj, res1

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in results]
    res = {}
    for src, (_, r) in zip(srcs, results):
        for k, v in r.items():
            res[k] = dict(v, file=src)
    pretty = demangle(sorted(res))
    report = {pretty[k][:400]: v for k, v in sorted(res.items())}
    bad = {k: v for k, v in report.items() if NO_SCRATCH.search(k) and (v.get("scratch", 0) or v.get("vgpr_spill", 0))}
    if bad:
        raise RuntimeError("kernels with counted vmcnt waits must not use scratch memory (a spill is a VMEM op and races with the "
                           "LDS-DMA ring):\n" + "\n".join(f"  {k}: {v}" for k, v in bad.items()))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(RESOURCES, "w") as f:
        json.dump(report, f, indent=0, sort_keys=True)
    with open(stamp, "w") as f:
        f.write(want)
    if verbose:
        spills = sum(1 for v in report.values() if v.get("scratch", 0))
        print(f"[icafusion_amd.build] built {LIB} from {len(srcs)} HIP sources for {ARCH}: {len(report)} kernels, "
              f"{spills} with scratch (none among the counted-vmcnt kernels)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
