"""CPU-only checks of the host side: config surface, module tree / state_dict layout, C-ABI export table, error paths."""
import ctypes
import os
import re

import pytest
import torch
import yaml

from helpers import REPO, load_cfg
from icafusion_amd import _lib, configs
from icafusion_amd.models.common import Conv, TransformerFusionBlock
from icafusion_amd.models.yolo import Model, parse_model


def test_library_exports_every_header_symbol():
    """libicaf.so loads and exports exactly the functions include/icaf.h declares (no compute calls: no GPU here)."""
    if not os.path.exists(_lib.LIB_PATH):
        from icafusion_amd.build import build
        build(verbose=False)
    header = open(os.path.join(REPO, "include", "icaf.h")).read()
    declared = set(re.findall(r"\b(icaf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name)
    assert handle.icaf_version() >= 100
    assert ctypes.sizeof(_lib.ConvArgs) == 312      # static_assert'ed on the C side (igemm.hip)
    assert ctypes.sizeof(_lib.BneckArgs) == 376


def test_yaml_generator_is_in_sync():
    for size, tag, nc in configs.VARIANTS:
        name = f"yolov5{size}_Transfusion_{tag}.yaml"
        assert load_cfg(name) == configs.transfusion_cfg(size, nc), name
    data = yaml.safe_load(open(os.path.join(REPO, "data", "multispectral", "kaist.yaml")))
    assert set(data) == {"path", "train_rgb", "val_rgb", "train_ir", "val_ir", "nc", "names"}


@pytest.mark.parametrize("name,params", [("yolov5s_Transfusion_kaist.yaml", 23261018)])
def test_model_construction_api(name, params):
    m = Model(os.path.join(REPO, "models", "transformer", name))
    assert sum(p.numel() for p in m.parameters()) == params
    assert len(m.model) == 38 and m.model[10].f == -4 and m.model[20].f == [4, 14]
    assert [type(l).__name__ for l in m.model][20:23] == ["TransformerFusionBlock"] * 3
    assert m.save == [4, 6, 6, 9, 14, 16, 19, 20, 21, 23, 27, 30, 33, 36]          # incl. the reference's duplicate 6
    assert m.stride.tolist() == [8.0, 16.0, 32.0] and m.names == ["0"]
    det = m.model[-1]
    assert (det.nl, det.na, det.no) == (3, 3, 6)
    assert torch.allclose(det.anchors[0] * 8, det.anchor_grid[0].view(3, 2))
    assert all(b.eps == 1e-3 for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))
    keys = set(m.state_dict())
    for k in ("model.0.conv.weight", "model.20.pos_emb_vis", "model.20.vis_coefficient.w1",
              "model.20.crosstransformer.0.crossatt.que_proj_vis.weight", "model.20.crosstransformer.0.ln_input.weight",
              "model.20.crosstransformer.0.mlp.2.bias", "model.20.crosstransformer.0.coefficient8.bias",
              "model.20.conv1x1_out.bn.running_var", "model.37.m.2.bias", "model.37.anchor_grid"):
        assert k in keys, k
    assert len(keys) == 728
    # overrides
    m2 = Model(load_cfg(name), nc=3)
    assert m2.model[-1].no == 8 and m2.yaml["nc"] == 3
    # drop-in module paths (pickled reference checkpoints resolve these)
    import models.common as mc
    import models.yolo_test as myt
    import models.yolo as my
    assert myt.Model is Model and my.Model is Model and mc.Conv is Conv and mc.TransformerFusionBlock is TransformerFusionBlock


def test_loops_argument_and_unsupported_modules():
    cfg = load_cfg("yolov5s_Transfusion_kaist_loops3.yaml")
    m = Model(cfg)
    assert all(m.model[i].crosstransformer[0].loops == 3 for i in (20, 21, 22))
    assert all(m.model[i].crosstransformer[0].crossatt.h == 8 for i in (20, 21, 22))
    # positional arguments keep the reference's meaning (models/yolo_test.py:284-286: args = [c2, *args[1:]]): a 4th positional is
    # `h`, the number of heads — NOT the iteration count
    cfg4 = load_cfg("yolov5s_Transfusion_kaist.yaml")
    cfg4["backbone"][20][3] = [256, 20, 20, 4]
    cfg4["backbone"][21][3] = [512, 16, 16, 4, 2, {"loops_num": 2}]
    m4 = Model(cfg4)
    assert m4.model[20].crosstransformer[0].crossatt.h == 4 and m4.model[20].crosstransformer[0].loops == 1
    assert m4.model[21].crosstransformer[0].crossatt.h == 4 and m4.model[21].crosstransformer[0].loops == 2
    assert m4.model[21].crosstransformer[0].mlp_vis[0].out_features == 2 * 256            # block_exp = 2
    with pytest.raises(ValueError, match="unknown yaml keyword"):
        bad_kw = load_cfg("yolov5s_Transfusion_kaist.yaml")
        bad_kw["backbone"][20][3] = [256, 20, 20, {"loops": 3}]
        Model(bad_kw)
    bad = load_cfg("yolov5s_Transfusion_kaist.yaml")
    bad["backbone"][1] = [-1, 1, "GhostConv", [128, 3, 2]]
    with pytest.raises(NotImplementedError):
        Model(bad)


def test_no_cpu_fallback_and_fuse():
    m = Model(load_cfg("yolov5s_Transfusion_kaist.yaml")).eval()
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(x, x)
    with pytest.raises(NotImplementedError):
        m.train()(x, x)
    m.fuse()
    assert not any(hasattr(c, "bn") for c in m.modules() if type(c) is Conv)
    assert "model.0.conv.bias" in m.state_dict()
    from icafusion_amd.utils.general import non_max_suppression
    with pytest.raises(RuntimeError):
        non_max_suppression(torch.zeros(1, 10, 6), 0.25, 0.45)


def test_layer_shapes_and_concat_placement():
    m = Model(load_cfg("yolov5s_Transfusion_kaist.yaml"))
    shapes = m._layer_shapes(1, 640, 640)
    assert shapes[0] == (32, 320, 320) and shapes[9] == (512, 20, 20) and shapes[20] == (128, 80, 80)
    assert shapes[25] == (512, 40, 40) and shapes[29] == (256, 80, 80) and shapes[36] == (512, 20, 20)


def test_host_metrics_match_reference_fixture():
    import numpy as np
    from helpers import load_golden
    from icafusion_amd.utils.metrics import ap_per_class
    from icafusion_amd.utils.general import box_iou, scale_coords
    g = load_golden("metrics_ap")
    r = ap_per_class(g["tp"], g["conf"], g["pcls"], g["tcls"])
    np.testing.assert_allclose(r[5], g["ap"], rtol=1e-9)
    np.testing.assert_allclose(r[3], g["p"], rtol=1e-9)
    np.testing.assert_allclose(r[4], g["r"], rtol=1e-9)
    np.testing.assert_allclose(box_iou(torch.from_numpy(g["box1"]), torch.from_numpy(g["box2"])).numpy(), g["iou"], rtol=1e-6)
    np.testing.assert_allclose(scale_coords((512, 640), torch.from_numpy(g["coords"]).clone(), (480, 720)).numpy(), g["scaled"], rtol=1e-6)


def test_pmc_summary_kernel_names_match_bench_names():
    """tools/pmc_summary.py maps rocprofv3 kernel names to the names bench.py reports (roofline.traffic is looked up by
    name: a template argument added to a kernel must not silently turn the traffic into null)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                               "tools", "pmc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = {
        "void icaf::igemm_dma_kernel<1, 1, 128, 128, 64, 32, 1, 128, 2, 1, false, false>(icaf::ConvP)": "igemm_dma128x2_bf16_bf16_128x128w8",
        "void icaf::igemm_dma_kernel<1, 1, 128, 128, 64, 64, 1, 64, 3, 2, false, true>(icaf::ConvP)": "igemm_dma64x3_bf16_bf16_128x128",
        "void icaf::igemm_dma_kernel<1, 0, 64, 64, 32, 32, 0, 64, 3, 1, false, false>(icaf::ConvP)": "igemm_dma64x3_bf16_f32_64x64",
        "void icaf::igemm_kernel<1, 1, 128, 64, 64, 32, 1>(icaf::ConvP)": "igemm_reg_bf16_bf16_128x64",
        "void icaf::ctile_kernel<1, 8, 32, 64, 1, 1>(icaf::ConvP, int, int, int, int, int)": "ctile_bf16_8x32n64",
        "void icaf::bneck_kernel<1, 8, 32, 32, 1, true>(icaf::ConvP, int, int, int, int, int)": "bottleneck+cv3",
        "void icaf::bneck_kernel<1, 8, 32, 32, 1, false>(icaf::ConvP, int, int, int, int, int)": "bottleneck",
        "void icaf::stem2_kernel<1, false>(icaf::Stem2P)": "stem+conv3x3s2+1x1",
        "void icaf::stem_kernel<1, 32, false>(icaf::StemP)": "stem",
        "void icaf::pool_tokens_rows_kernel<1, 12>(icaf::Elem<1>::type const*, int)": "dmff_pool_tokens",
        "void icaf::detect_pixel_kernel<3, 6>(float const*, int, float*)": "detect_decode",
        "void icaf::igemm_stream_kernel<1, 128, 1>(icaf::ConvP)": "igemm_stream_bf16_128x128",
        "void icaf::igemm_stream_kernel<2, 64, 0>(icaf::ConvP)": "igemm_stream_f16_128x64",
        "void icaf::igemm_wreg_kernel<1, 8, 1, 2>(icaf::ConvP, void const*, long long)": "igemm_wreg_bf16_128x256",
        "void icaf::igemm_wreg_kernel<1, 8, 1, 2, 1, 128>(icaf::ConvP, void const*, long long)": "igemm_wreg_bf16_128x256",
        "void icaf::igemm_wreg_kernel<1, 4, 1, 2, 2, 128>(icaf::ConvP, void const*, long long)": "igemm_wreg_bf16_128x256w4",
        "void icaf::igemm_wreg_kernel<2, 4, 0, 1, 1, 64>(icaf::ConvP, void const*, long long)": "igemm_wreg_f16_64x128",
        "void icaf::igemm_wreg_kernel<1, 4, 1, 1, 2, 64>(icaf::ConvP, void const*, long long)": "igemm_wreg_bf16_64x256",
        "void icaf::igemm_wreg_kernel<1, 8, 1, 1, 2, 128>(icaf::ConvP, void const*, long long)": "igemm_wreg_bf16_128x512",
        "void icaf::detect_conv_kernel<1, 3, 6>(icaf::ConvP, icaf::DetectEpi<3, 6>)": "detect_conv+decode",
        "void icaf::detect_decode_kernel<true>(float const*, int, float*)": "detect_decode",
        "void icaf::upsample_kernel<true>(unsigned int __vector(4) const*, int)": "upsample_nearest",
        "void icaf::pool_tokens_kernel<1>(icaf::Elem<1>::type const*, int)": "dmff_pool_tokens",
        "void icaf::sppf_lds_kernel<1>(icaf::Elem<1>::type const*, int)": "sppf_pool",
        "void icaf::dmff_attn_mlp_kernel<1, 16, 1, 128>(icaf::DmffP)": "dmff_attn_mlp",
        "void icaf::dmff_ln_qkv_kernel<1, 128>(icaf::DmffP)": "dmff_ln_qkv",
        "void icaf::cross_attn_kernel<1, 32>(icaf::Elem<1>::type const*, icaf::Elem<1>::type*, int, int, int, int, int, float)": "cross_attention",
        "icaf::nms_walk_kernel(float const*, long long, int)": "nms_walk_kernel",
        # round 3: the PRE flag added to igemm_stream_kernel once merged its two tiles under one name — every template tail must be tolerated
        "void icaf::igemm_stream_kernel<1, 128, 1, 1, false>(icaf::ConvP)": "igemm_stream_bf16_128x128",
        "void icaf::igemm_stream_kernel<1, 64, 1, 2, true>(icaf::ConvP)": "igemm_stream_bf16_128x64",
        "void icaf::stem2_kernel<1, false, true>(icaf::Stem2P)": "stem+conv3x3s2+1x1",
        "void icaf::cstream_kernel<1, true>(icaf::ConvP, icaf::CsGeom)": "cstream_bf16_8x16n64",
        "void icaf::cwide_kernel<1, 128, 1, 4, false>(icaf::ConvP, icaf::CwGeom, void const*, long long)": "cwide_bf16_8x16n128",
        "void icaf::cwide_kernel<1, 128, 1, 2, true>(icaf::ConvP, icaf::CwGeom, void const*, long long)": "cwide_bf16_8x8n128",
        "void icaf::cwide_kernel<2, 64, 2, 2, true>(icaf::ConvP, icaf::CwGeom, void const*, long long)": "cwide_f16_8x8n128s2c64",
        "void icaf::cwide_kernel<1, 128, 2, 2, false>(icaf::ConvP, icaf::CwGeom, void const*, long long)": "cwide_bf16_8x8n128s2",
        "void icaf::dmff_wide_ln_qkv_kernel<1>(icaf::WideP)": "dmff_ln_qkv",
        "void icaf::dmff_wide_proj_mlp_kernel<2, 2>(icaf::WideP)": "dmff_proj_mlp",
    }
    for raw, want in cases.items():
        assert mod.short(raw) == want, raw


def test_plan_cache_is_a_byte_capped_lru():
    """Model.plan_for keeps plans in least-recently-used order under `plan_cache_bytes` (a rectangular-batch validation run
    meets many (B, H, W); each plan pins all of its intermediates).  CPU: plans build without a GPU."""
    m = Model(load_cfg("yolov5n_Transfusion_kaist.yaml")).eval()
    shapes = [(1, 320, 320), (1, 320, 352), (1, 352, 320), (2, 320, 320), (1, 384, 320), (1, 320, 384)]
    one = m.plan_for(*shapes[0], device="cpu", dtype=torch.bfloat16)
    assert one.nbytes > 0 and m.plan_for(*shapes[0], device="cpu", dtype=torch.bfloat16) is one           # cache hit
    m.plan_cache_bytes = int(3.5 * one.nbytes)
    for s in shapes:
        m.plan_for(*s, device="cpu", dtype=torch.bfloat16)
        total = sum(p.nbytes for p in m._plans.values())
        assert total <= m.plan_cache_bytes and len(m._plans) <= 3
    keys = [k[:3] for k in m._plans]
    assert keys[-1] == shapes[-1] and shapes[0] not in keys                                                # oldest evicted, newest kept
    m.plan_for(*keys[0], device="cpu", dtype=torch.bfloat16)                                               # touch the oldest survivor ...
    m.plan_for(1, 416, 320, device="cpu", dtype=torch.bfloat16)
    assert keys[0] in [k[:3] for k in m._plans] and keys[1] not in [k[:3] for k in m._plans]               # ... so the next one goes
    m.plan_cache_bytes = 1                                                                                 # the newest plan always stays
    assert m.plan_for(1, 448, 320, device="cpu", dtype=torch.bfloat16) is not None and len(m._plans) == 1


def test_no_counted_vmcnt_kernel_uses_scratch():
    """icafusion_amd/build.py records every kernel's resource usage; kernels that order their LDS-DMA ring with counted
    `s_waitcnt vmcnt(N)` must have zero scratch (a spill is a VMEM op on the same counter — ADVICE r1), and the build refuses
    to link otherwise.  Here: the report of the library that is actually loaded says so."""
    import json
    import os
    from icafusion_amd import build
    assert os.path.exists(build.RESOURCES), "run `python -m icafusion_amd.build`"
    rep = json.load(open(build.RESOURCES))
    guarded = {k: v for k, v in rep.items() if build.NO_SCRATCH.search(k)}
    assert len(guarded) > 100 and len(rep) > len(guarded)
    assert all(v.get("scratch", 0) == 0 and v.get("vgpr_spill", 0) == 0 for v in guarded.values())
    assert all(v["vgpr"] + v.get("agpr", 0) <= 512 for v in rep.values())


def test_fastdiv_matches_integer_division(tmp_path):
    """icaf::FastDiv (multiply-high division used by the element-per-thread kernels' index math) against `/` on the host:
    tests/native/fastdiv_check.cpp includes the kernels' own header and sweeps divisors x the 31-bit range."""
    import subprocess

    from icafusion_amd.build import hipcc
    exe = tmp_path / "fastdiv_check"
    r = subprocess.run([hipcc(), "-O2", "-x", "hip", "--cuda-host-only", "-I", os.path.join(REPO, "icafusion_amd", "csrc"),
                        os.path.join(REPO, "tests", "native", "fastdiv_check.cpp"), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fastdiv ok" in r.stdout, r.stdout + r.stderr


def test_committed_tune_caches_only_name_configurations_the_tuner_would_time():
    """Every (signature -> launch configuration) pair in profiles/tune_cache*.json is one conv_candidates() offers for that
    signature: a cache edited by hand, or left behind by a change of the candidate rules, cannot pin a configuration the
    library rejects or the bit-identity tests never see."""
    import glob
    import json
    from types import SimpleNamespace

    from icafusion_amd import ops
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "tune_cache*.json")))
    assert files
    n = 0
    for f in files:
        for key, tile in json.load(open(f)):
            (M, cout, cin, kh, kw, sh, sw, H, W, ldx, ldy, groups, dtype, out_dtype, act, res, pre, cout2, chain_keep) = key
            cw = act == ops.ACT_SILU and bool(ops.cwide_shapes(kh, kw, sh, sw, kh // 2, kw // 2, cin, cout))
            wf = (dtype != ops.F32 and out_dtype == dtype and (cin * 2) % 128 == 0 and not pre and ((not cout2 and cout > 64) or cw))   # ops.conv2d's rule
            a = SimpleNamespace(Cout=cout, Cin=cin, kh=kh, kw=kw, sh=sh, sw=sw, ph=kh // 2, pw=kw // 2, dtype=dtype, out_dtype=out_dtype,
                                act=act, pre=bool(pre), w2=bool(cout2), Cout2=cout2, res=bool(res), wf=wf, groups=groups, B=1, Ho=1, Wo=M,
                                x2=chain_keep == 2)        # (the last field is 2 for a C3 tail: ops._conv_signature)
            assert tile in ops.conv_candidates(a), (os.path.basename(f), key, tile)
            assert ldy >= cout and ldx >= cin and M > 0 and groups in (1, 2)
            n += 1
    assert n > 150


def test_wide_dmff_entry_points_reject_what_they_are_not_built_for():
    """icaf_dmff_wide_*: argument checks run before any device call — a shape outside C in {256, 512}, fp32, or a hidden width that is
    not a multiple of 256 is an error with a reason, never a silent fall-back (callers then keep the per-layer launches)."""
    import ctypes as C

    from icafusion_amd import _lib
    from icafusion_amd._lib import DmffArgs, lib
    a = DmffArgs()
    buf = (C.c_char * 4096)()
    ptr = C.cast(buf, C.c_void_p).value
    for f in ("x", "qkv", "y", "wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2", "ln_mlp_gamma", "ln_mlp_beta"):
        setattr(a, f, ptr)
    for i in range(2):
        a.ln_attn_gamma[i] = ptr; a.ln_attn_beta[i] = ptr
    a.dtype, a.B, a.N, a.heads, a.Kp, a.Kp4, a.hidden, a.ldy = 1, 2, 16, 8, 384, 1536, 1536, 384
    a.C = 384
    assert lib().icaf_dmff_wide_ln_qkv(C.byref(a), None) != 0 and b"C=384" in lib().icaf_last_error()
    assert lib().icaf_dmff_wide_proj_mlp(C.byref(a), ptr, None) != 0
    a.C, a.Kp, a.Kp4, a.hidden, a.ldy, a.dtype = 256, 256, 1024, 1024, 256, 0          # fp32
    assert lib().icaf_dmff_wide_proj_mlp(C.byref(a), ptr, None) != 0 and b"16-bit" in lib().icaf_last_error()
    a.dtype, a.hidden, a.Kp4 = 1, 1152, 1152                                             # hidden % 256 != 0
    assert lib().icaf_dmff_wide_proj_mlp(C.byref(a), ptr, None) != 0 and b"256" in lib().icaf_last_error()
    a.hidden, a.Kp4 = 1024, 1024
    assert lib().icaf_dmff_wide_proj_mlp(C.byref(a), None, None) != 0                    # attention output missing
    assert _lib.IcafError


def test_isa_mix_splits_kernels_and_classifies_instructions(tmp_path):
    """tools/isa_mix.py (the static instruction-mix table under profiles/): a kernel runs from its label to .Lfunc_end — a uniform early exit puts a
    second s_endpgm in the middle — and every instruction lands in one class."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_mix", os.path.join(REPO, "tools", "isa_mix.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    asm = tmp_path / "k.s"
    asm.write_text("\n".join([
        "\t.text", "helper:", "\tv_mov_b32_e32 v0, 0", "\ts_setpc_b64 s[30:31]", ".Lfunc_end0:",
        "k1: ; @k1", "\ts_load_dwordx2 s[0:1], s[4:5], 0x0", "\ts_waitcnt lgkmcnt(0)", "\ts_cbranch_scc1 .LBB1_2", "\tv_exp_f32_e32 v1, v0", "\ts_endpgm",
        ".LBB1_2:", "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], 0", "\tds_read_b128 v[0:3], v4", "\tglobal_store_dwordx4 v[0:1], v[2:5], off",
        "\ts_barrier", "\tv_rcp_f32_e32 v2, v1", "\tv_add_f32_e32 v3, v2, v1", "\ts_endpgm", ".Lfunc_end1:",
        "\t.amdhsa_kernel k1", "\t.end_amdhsa_kernel"]))
    got = m.kernels_of(str(asm))
    assert list(got) == ["k1"]
    assert dict(got["k1"]) == {"salu": 4, "waitcnt": 1, "trans": 2, "mfma": 1, "lds": 1, "vmem": 1, "barrier": 1, "valu": 1}


def test_rank_cpu_affinity_follows_the_gpus_numa_node(tmp_path):
    """dist.pin_to_gpu_numa (bench.py, N > 1): the CPUs of the GPU's NUMA node as sysfs reports them; equal contiguous slices of the allowed CPUs
    when the platform does not say.  Runs in a child process: the affinity of the test runner is left alone."""
    import subprocess
    import sys
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs")
    half = allowed[:len(allowed) // 2]
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text(f"{half[0]}-{half[-1]}\n")
    unknown = tmp_path / "bus" / "pci" / "devices" / "0000:c2:00.0"
    unknown.mkdir(parents=True)
    (unknown / "numa_node").write_text("-1\n")
    code = ("import json, os, sys; sys.path.insert(0, %r); from icafusion_amd import dist as D\n"
            "a = D.pin_to_gpu_numa(0, 2, 'C1:00.0', sysfs=%r); ca = sorted(os.sched_getaffinity(0))\n"
            "os.sched_setaffinity(0, %r)\n"
            "b = D.pin_to_gpu_numa(1, 2, '0000:c2:00.0', sysfs=%r); cb = sorted(os.sched_getaffinity(0))\n"
            "print(json.dumps([a, ca, b, cb]))") % (REPO, str(tmp_path), allowed, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    import json
    a, ca, b, cb = json.loads(r.stdout.strip().splitlines()[-1])
    assert a["pinned"] and a["numa_node"] == 1 and ca == [c for c in half if c in allowed]
    assert b["pinned"] and b["numa_node"] is None and cb == allowed[len(allowed) // 2:]


def test_plan_options_are_one_object_filled_from_one_variable():
    """options.PlanOptions: every execution switch is a field; ICAF_OPTIONS (field=value, comma-separated) fills them, the legacy per-switch variables
    still work, an unknown name is an error (a misspelt switch must not silently run the default), and the library's probe knobs go through
    icaf_set_option — the C side reads no environment variable."""
    from icafusion_amd.options import PlanOptions
    d = PlanOptions()
    assert d.non_default() == {} and d.dmff_fuse and d.retune_tiles == frozenset()
    o = PlanOptions.from_env({"ICAF_OPTIONS": "dmff_fuse=0,retune_tiles=63:64, pipe_copy_prio=0", "ICAF_DMFF_WIDE": "0", "ICAF_RETUNE_TILES": "61,62"})
    assert o.non_default() == {"dmff_fuse": False, "dmff_wide": False, "retune_tiles": [63, 64], "pipe_copy_prio": 0}      # ICAF_OPTIONS wins over a legacy name
    with pytest.raises(ValueError):
        PlanOptions.from_env({"ICAF_OPTIONS": "dmff_fuze=0"})
    assert set(o.lib_options()) == {"detect_elementwise", "attn_qsplit", "sppf_vpb"}
    l = _lib.lib()
    assert l.icaf_set_option(b"attn_qsplit", 0) == 0 and l.icaf_set_option(b"no_such_knob", 1) != 0 and b"unknown option" in l.icaf_last_error()
    src = os.path.join(REPO, "icafusion_amd")
    for root, _, files in os.walk(src):                      # one place: no other module of the package reads an ICAF_* switch from the environment
        for f in files:
            if f.endswith((".py", ".hip", ".h")) and f not in ("options.py", "build.py", "_lib.py"):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"environ[^\n]*ICAF_|getenv\(\"ICAF_(?!S2_CLK)", text), f
