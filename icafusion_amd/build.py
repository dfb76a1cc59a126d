"""Build libicaf.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot to the GPU box).

    python -m icafusion_amd.build [--force]

One object per .hip file (parallel), then one shared library exporting the C ABI of include/icaf.h.
detect.hip / nms.hip are built with fp contraction off: their arithmetic must round like the reference's
separate fp32 torch ops so that NMS keep-indices are bit-exact.
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "_obj")
LIB = os.path.join(LIBDIR, "libicaf.so")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-comment"]
PER_FILE = {"detect.hip": ["-ffp-contract=off"], "nms.hip": ["-ffp-contract=off"]}
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


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "icaf.h"))
    srcs = sources()
    stamp = LIB[:-3] + ".stamp"
    want = _digest([os.path.join(CSRC, s) for s in srcs] + headers, " ".join(COMMON) + repr(sorted(PER_FILE.items())))
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
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(want)
    if verbose:
        print(f"[icafusion_amd.build] built {LIB} from {len(srcs)} HIP sources for {ARCH}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
