"""ONE place for the execution switches of the plan builder, the tuner and the serving pipeline.

Every switch that steers WHICH kernels a plan is built from (never what they compute: every configuration of a layer produces the same
bits, tests/test_gpu_fullsize.py) is a field of `PlanOptions`; the process-wide instance is `OPT`.  It is filled once, at import, from

    ICAF_OPTIONS="dmff_fuse=0,retune_tiles=63:64,pipe_copy_prio=0"      (comma-separated field=value; sets use ':')

and, for the A/B scripts of earlier rounds, from the legacy per-switch variable each field names (`ICAF_DMFF_FUSE=0` ...).  bench.py logs
`OPT.as_dict()` and `OPT.non_default()` into its JSON line, so a number can always be traced to the switches it ran with.  The modules of
models/common.py keep their CLASS-level defaults (a reference checkpoint un-pickles without running our constructors): those defaults are
read from `OPT` when the class body executes.  Library-side probe knobs (fields tagged lib=True) are pushed into libicaf.so through
`icaf_set_option` when the library is loaded — the C side reads no environment variable.
"""
import dataclasses
import os
from dataclasses import dataclass, field


def _f(default, env, doc, lib=False):
    return field(default=default, metadata={"env": env, "doc": doc, "lib": lib})


@dataclass
class PlanOptions:
    # ---- tuner candidates (ops.conv_candidates) -------------------------------------------------------------------------------
    cwide: bool = _f(True, "ICAF_CWIDE", "resident-patch / streamed-weights 3x3 kernels (cwide.hip) as tuner candidates")
    cstream: bool = _f(True, "ICAF_CSTREAM", "persistent resident-filter 3x3 kernel (cstream.hip) as a tuner candidate")
    wreg_gemm: bool = _f(True, "ICAF_WREG_GEMM", "weights-from-registers GEMM kernels (igemm_wreg.hip) as tuner candidates")
    stream_gemm: bool = _f(True, "ICAF_STREAM_GEMM", "persistent streaming 1x1 kernel (igemm_stream.hip) as a tuner candidate")
    wreg64_maxpix: int = _f(128 * 1024, "ICAF_WREG64_MAXPIX", "launches with at most this many pixels are offered the 64-pixel wreg tiles")
    tail_8x16_minpix: int = _f(200_000, "ICAF_TAIL_8X16_MINPIX", "C3 tails with fewer pixels per stream are offered only the 8 x 8 cwide form")
    retune_tiles: frozenset = _f(frozenset(), "ICAF_RETUNE_TILES", "launch configurations that get their chance against a cached choice (e.g. 63:64)")
    retune_pre: bool = _f(False, "ICAF_RETUNE_PRE", "every configuration of the pre-activation-term launches is re-timed against the cached one")
    # ---- plan structure: class defaults of the host mirror (models/common.py) ---------------------------------------------------
    c3_tail: bool = _f(True, "ICAF_C3_TAIL", "a C3's cv3 rides on its last Bottleneck's 3x3 launch")
    dmff_fuse: bool = _f(True, "ICAF_DMFF_FUSE", "fused DMFF block launches (two / three per iteration) instead of seven per-layer launches")
    dmff_fuse_max_c: int = _f(64, "ICAF_DMFF_FUSE_MAX_C", "largest C the two-launch block kernel (dmff_fused.hip) takes")
    dmff_fuse_fp32: bool = _f(False, "ICAF_DMFF_FUSE_FP32", "fp32 plans also use the fused block kernels (their fp32 instantiations)")
    dmff_wide: bool = _f(True, "ICAF_DMFF_WIDE", "three-launch block kernels (dmff_wide.hip)")
    dmff_wide_max_c: int = _f(512, "ICAF_DMFF_WIDE_MAX_C", "largest C the three-launch block kernels take")
    dmff_res32: bool = _f(True, "ICAF_DMFF_RES32", "loops > 1: the token stream between iterations in fp32")
    detect_fuse: bool = _f(True, "ICAF_DETECT_FUSE", "Detect's 1x1 conv and the decode in one launch")
    dmff_qkv_npass: int = _f(0, "ICAF_DMFF_QKV_NPASS", "output-channel passes per workgroup of the wide LN + QKV kernel (0 = automatic)")
    dmff_ksplit: int = _f(0, "ICAF_DMFF_KSPLIT", "hidden-column split of the wide out-proj + MLP kernel (0 = automatic, 1 = never, 2 / 4 = force)")
    # ---- serving pipeline (pipeline.py) -----------------------------------------------------------------------------------------
    pipe_extra_plans: int = _f(1, "ICAF_PIPE_EXTRA_PLANS", "further sets of `depth` plans a host-fed pipeline owns (copy targets not in flight)")
    pipe_copy_streams: int = _f(1, "ICAF_PIPE_COPY_STREAMS", "a batch's host -> device copy in this many slices, one copy stream each")
    pipe_copy_prio: int = _f(-1, "ICAF_PIPE_COPY_PRIO", "priority of the copy stream(s): -1 = high (a hardware queue of their own)")
    pipe_branches: bool = _f(False, "ICAF_PIPE_BRANCHES", "host-fed pipelines keep the hipGraph's parallel branches (measured slower)")
    # ---- library-side probe knobs (pushed through icaf_set_option) ----------------------------------------------------------------
    detect_elementwise: bool = _f(False, "ICAF_DETECT_ELEMENTWISE", "Detect decode by the one-thread-per-element kernel", lib=True)
    attn_qsplit: int = _f(0, "ICAF_ATTN_QSPLIT", "query splits per head of the attention kernel (0 = automatic)", lib=True)
    sppf_vpb: int = _f(0, "ICAF_SPPF_VPB", "channel vectors per workgroup of the SPPF kernel (0 = automatic)", lib=True)

    @staticmethod
    def _parse(f, text):
        t = f.type if isinstance(f.type, type) else {"bool": bool, "int": int, "frozenset": frozenset}[str(f.type)]
        if t is bool:
            return text.strip().lower() not in ("0", "", "false", "no", "off")
        if t is int:
            return int(text)
        if t is frozenset:
            return frozenset(int(x) for x in text.replace(",", ":").split(":") if x.strip())
        raise TypeError(f.name)

    @classmethod
    def from_env(cls, env=None):
        env = os.environ if env is None else env
        o = cls()
        fields = {f.name: f for f in dataclasses.fields(cls)}
        for f in fields.values():                                        # legacy per-switch variables
            if f.metadata["env"] in env:
                setattr(o, f.name, cls._parse(f, env[f.metadata["env"]]))
        for item in env.get("ICAF_OPTIONS", "").split(","):              # the one variable: wins over the legacy names
            if not item.strip():
                continue
            k, sep, v = item.partition("=")
            k = k.strip()
            if k not in fields or not sep:
                raise ValueError(f"ICAF_OPTIONS: unknown switch {item!r} (known: {', '.join(sorted(fields))})")
            setattr(o, k, cls._parse(fields[k], v))
        return o

    def as_dict(self):
        return {f.name: (sorted(getattr(self, f.name)) if isinstance(getattr(self, f.name), frozenset) else getattr(self, f.name))
                for f in dataclasses.fields(self)}

    def non_default(self):
        d = PlanOptions()
        return {k: v for k, v in self.as_dict().items() if v != d.as_dict()[k]}

    def lib_options(self):
        return {f.name: int(getattr(self, f.name)) for f in dataclasses.fields(self) if f.metadata["lib"]}


OPT = PlanOptions.from_env()
