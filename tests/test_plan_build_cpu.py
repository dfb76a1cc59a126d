"""Host logic without a GPU: the execution plan of every shipped configuration family builds on the CPU (weight packing,
buffer placement, launch records — no kernel is called), and the plan-level fusions appear exactly where their
conditions hold."""
import pytest
import torch

from helpers import load_cfg
from icafusion_amd.models.common import C3, Conv
from icafusion_amd.models.yolo import Model

CONFIGS = ["yolov5n_Transfusion_kaist.yaml", "yolov5s_Transfusion_kaist.yaml", "yolov5m_Transfusion_kaist.yaml",
           "yolov5l_Transfusion_VEDAI.yaml", "yolov5s_Add_kaist.yaml", "yolov5n_NiNfusion_FLIR.yaml", "yolov5m_NiNfusion_kaist.yaml"]


def names(plan):
    return [l.name for l in plan.launches]


@pytest.mark.parametrize("cfg_name", CONFIGS)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_plan_builds_on_cpu(cfg_name, dtype):
    m = Model(load_cfg(cfg_name)).eval()
    plan = m.build_plan(2, 320, 352, "cpu", dtype)
    n = names(plan)
    assert len(n) > 40 and n[-1] == ("detect_decode" if dtype == torch.float32 else "detect_conv+decode")    # 16-bit: a Detect level is one launch
    assert all(l.flops >= 0 and l.bytes >= 0 for l in plan.launches)
    if dtype == torch.float32:                      # the fused / persistent kernels are 16-bit only
        assert not any(x.startswith("stem") or x.startswith("bottleneck") or x.endswith("+1x1") for x in n)


def test_yolov5s_plan_level_fusions_and_their_switches():
    m = Model(load_cfg("yolov5s_Transfusion_kaist.yaml")).eval()
    base = names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))
    assert base[0] == "stem+conv3x3s2+1x1" and base[1] == "bottleneck+cv3" and base.count("conv3x3s1+1x1") == 3
    assert "upsample_nearest" in base and "c3_up_term" not in base           # folded up-sampling is off by default
    assert base.count("conv3x3s1+cv3") == 3 and len(base) == 63              # the three C3 blocks with 128-channel Bottlenecks end in ONE launch (3x3 + cv3)
    u8 = names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16, u8=True))
    assert u8 == base                                                        # same launches from the uint8 batch
    try:
        Conv.fuse_stem2 = False
        assert names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))[:2] == ["stem", "conv3x3s2+1x1"]
        Conv.fuse_stem2, C3.fuse_cv3 = True, False
        assert names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))[1:3] == ["bottleneck", "conv1x1s1"]
        C3.fuse_cv3, C3.chain_bottlenecks = True, False
        assert "conv3x3s1+1x1" not in names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))
        C3.chain_bottlenecks, Conv.chain_tail = True, False
        plain = names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))
        assert "conv3x3s1+cv3" not in plain and len(plain) == len(base) + 3
    finally:
        Conv.fuse_stem2, C3.fuse_cv3, C3.chain_bottlenecks, Conv.chain_tail = True, True, True, True
    m.fold_upsample = True
    folded = names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))
    assert folded.count("c3_up_term") == 2 and "upsample_nearest" not in folded and len(folded) == len(base)
    m.pair_streams = False
    m.fold_upsample = False
    assert len(names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))) > len(base) + 20     # one launch per stream and layer


def test_wider_models_keep_the_generic_launches():
    """icaf_stem2 / the chained cv3 are built for the yolov5s widths only; other widths must fall back, not fail."""
    for cfg_name in ("yolov5n_Transfusion_kaist.yaml", "yolov5l_Transfusion_VEDAI.yaml"):
        n = names(Model(load_cfg(cfg_name)).eval().build_plan(1, 320, 320, "cpu", torch.bfloat16))
        assert "stem+conv3x3s2+1x1" not in n and "bottleneck+cv3" not in n
        # the C3 tail follows the Bottleneck width (c_ = 128): yolov5l has it at P3, yolov5n at P5 — backbone and head block each
        assert n.count("conv3x3s1+cv3") == 2


def _dmff_launches(n):
    """launch names of the three DMFF blocks of a plan, split at the token pooling that opens each block"""
    idx = [i for i, x in enumerate(n) if x == "dmff_pool_tokens"]
    blocks = []
    for j, i in enumerate(idx):
        end = next(k for k in range(i + 1, len(n)) if n[k].startswith("dmff_tail") or n[k] == "dmff_upsample_merge")
        blocks.append(n[i + 1:end])
    return blocks


@pytest.mark.parametrize("loops", [1, 3])
def test_dmff_block_launch_structure_at_every_level(loops):
    """16-bit yolov5s: every level runs LN + QKV, attention, out-projection + LN + MLP (3 launches per iteration; CrossTransformerBlock.fuse_max_c
    = 128 gives P3 the two-launch form of dmff_fused.hip back: LN + QKV, attention + out-projection + LN + MLP) — VERDICT r2 #3; P5, whose few 64-row tiles leave most CUs idle
    and whose weights overflow an XCD's L2, splits the MLP's hidden columns over several workgroups per tile and adds a small
    reduce launch (VERDICT r3 #1c).  fp32 keeps the seven per-layer launches."""
    m = Model(load_cfg("yolov5s_Transfusion_kaist.yaml")).eval()
    for i in (20, 21, 22):
        m.model[i].crosstransformer[0].loops = loops
    blocks = _dmff_launches(names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16)))
    assert blocks[0] == ["dmff_ln_qkv", "cross_attention", "dmff_proj_mlp"] * loops          # (round 4: the four-wavefront build at C = 128)
    assert blocks[1] == ["dmff_ln_qkv", "cross_attention", "dmff_proj_mlp"] * loops
    assert blocks[2] == ["dmff_ln_qkv", "cross_attention", "dmff_proj_mlp", "dmff_proj_mlp_reduce"] * loops
    from icafusion_amd import ops
    assert ops.dmff_wide_ksplit(100, 512, 2048) == 2                                                      # P5 of yolov5s — at every batch size
    assert ops.dmff_wide_ksplit(256, 256, 1024) == 1 and ops.dmff_wide_ksplit(256, 512, 2048) == 1        # P4 of yolov5s / yolov5l
    from icafusion_amd.models.common import CrossTransformerBlock
    try:
        CrossTransformerBlock.fuse_wide = False                  # A/B switch: the wide levels fall back to the per-layer launches
        plain = _dmff_launches(names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16)))
        assert len(plain[0]) == 7 * loops and len(plain[1]) == 7 * loops and len(plain[2]) == 7 * loops
        CrossTransformerBlock.fuse_wide, CrossTransformerBlock.fuse_max_c = True, 128          # the two-launch form at P3
        two = _dmff_launches(names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16)))
        assert two[0] == ["dmff_ln_qkv", "dmff_attn_mlp"] * loops and two[1] == blocks[1]
    finally:
        CrossTransformerBlock.fuse_wide, CrossTransformerBlock.fuse_max_c = True, 64
    f32 = _dmff_launches(names(m.build_plan(2, 320, 320, "cpu", torch.float32)))
    assert [len(b) for b in f32] == [7 * loops] * 3


def test_fragment_major_weight_copy_layout():
    """ops.frag_weights: [G][Np/32][Kp/16][64][8] with lane (hi * 32 + r) of block (nb, ks) = w[nb * 32 + r][ks * 16 + hi * 8 : + 8] —
    the operand one 16-byte load per lane hands to v_mfma_f32_32x32x16 (icaf.h: icaf_conv_args.wf, the dmff_wide entry points)."""
    from icafusion_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(2, 128, 96, generator=g).to(torch.bfloat16)
    f = ops.frag_weights(w)
    assert f.shape == (2, 4, 6, 64, 8) and f.dtype == w.dtype and f.stride(0) == w.stride(0)
    for (gi, nb, ks, lane) in ((0, 0, 0, 0), (1, 3, 5, 63), (0, 2, 4, 37), (1, 1, 2, 31), (0, 3, 0, 32)):
        hi, r = lane >> 5, lane & 31
        assert torch.equal(f[gi, nb, ks, lane], w[gi, nb * 32 + r, ks * 16 + hi * 8: ks * 16 + hi * 8 + 8])
    assert ops.frag_weights(w) is f                               # built once per packed tensor
    w2 = torch.randn(64, 32, generator=g).to(torch.float16)
    assert ops.frag_weights(w2).shape == (2, 2, 64, 8)


def test_resident_patch_kernel_shape_table():
    """ops.cwide_shapes: which cwide.hip launch configurations exist for a 3x3 layer (the tuner only offers these)."""
    from icafusion_amd import ops
    s = ops.cwide_shapes
    assert s(3, 3, 1, 1, 1, 1, 128, 128) == [81, 82]
    assert s(3, 3, 2, 2, 1, 1, 64, 128) == [83, 85] and s(3, 3, 2, 2, 1, 1, 64, 256) == [83, 85]
    assert s(3, 3, 2, 2, 1, 1, 128, 128) == [84] and s(3, 3, 2, 2, 1, 1, 128, 256) == [84]
    for bad in ((1, 1, 1, 1, 0, 0, 128, 128), (3, 3, 1, 1, 1, 1, 128, 64), (3, 3, 1, 1, 1, 1, 256, 256), (3, 3, 1, 1, 0, 0, 128, 128),
                (3, 3, 2, 1, 1, 1, 64, 128), (3, 3, 2, 2, 1, 1, 32, 128), (3, 3, 1, 1, 1, 1, 64, 128)):
        assert s(*bad) == []
    assert ops.dmff_wide_ok(256, 1024, torch.bfloat16) and ops.dmff_wide_ok(512, 2048, torch.float16)
    assert ops.dmff_wide_ok(128, 512, torch.bfloat16) and not ops.dmff_wide_ok(384, 1536, torch.bfloat16) and not ops.dmff_wide_ok(1024, 4096, torch.bfloat16)
    assert not ops.dmff_wide_ok(256, 1024, torch.float32) and not ops.dmff_wide_ok(512, 2048 + 128, torch.bfloat16)


def test_c3_tail_launch_configurations_follow_the_map_size():
    """ops.conv_candidates for a C3 tail (cv3 chained behind the last Bottleneck's 3x3, icaf_conv_args.x2): only cwide.hip's two forms, and below
    OPT.tail_8x16_minpix pixels per stream only the 8 x 8 form (the 8 x 16 form wins isolated timings there and loses the bench: docs/HISTORY.md section 15);
    the tuner signature keeps tails apart from chain_keep launches of the same shape."""
    from types import SimpleNamespace

    from icafusion_amd import ops

    def tail(m):
        return SimpleNamespace(Cout=128, Cin=128, kh=3, kw=3, sh=1, sw=1, ph=1, pw=1, dtype=ops.BF16, out_dtype=ops.BF16, act=ops.ACT_SILU, pre=False,
                               w2=True, x2=True, Cout2=256, res=True, wf=True, groups=2, B=1, Ho=1, Wo=m, H=1, W=m, ldx=128, ldy=256, pre_mode=0,
                               chain_keep=0)
    assert ops.conv_candidates(tail(51200)) == [82] and ops.conv_candidates(tail(81920)) == [82]
    assert ops.conv_candidates(tail(204800)) == [81, 82] and ops.conv_candidates(tail(409600)) == [81, 82]
    a = tail(51200)
    b = SimpleNamespace(**{**vars(a), "x2": False, "chain_keep": 1})
    assert ops._conv_signature(a) != ops._conv_signature(b) and ops._conv_signature(a)[-1] == 2


def test_fusion_report_shows_which_width_specialised_launches_a_plan_runs():
    """Plan.fusion_report() (bench.py: config.fused_paths): yolov5s runs the fused stem, the fused Bottleneck + cv3 and the three-launch DMFF block
    at every level; yolov5l's widths fall outside icaf_stem2 / icaf_bottleneck and its P5 level (C = 1024) outside the three-launch form — the
    report says so instead of the slower launches running silently."""
    rs = Model(load_cfg("yolov5s_Transfusion_kaist.yaml")).eval().build_plan(2, 320, 320, "cpu", torch.bfloat16).fusion_report()
    assert rs["launches"] == 63 and rs["stem"].startswith("stem2") and rs["bottleneck_fused"] == 1 and rs["dmff_blocks"] == "three-launch"
    assert rs["dmff_levels_three_launch"] == 3 and rs["dmff_levels_per_layer"] == 0 and rs["detect"].startswith("conv+decode")
    rl = Model(load_cfg("yolov5l_Transfusion_VEDAI.yaml")).eval().build_plan(1, 320, 320, "cpu", torch.float16).fusion_report()
    assert rl["stem"] == "stem" and rl["bottleneck_fused"] == 0 and rl["dmff_levels_three_launch"] == 2 and rl["dmff_levels_per_layer"] == 1
    r32 = Model(load_cfg("yolov5s_Transfusion_kaist.yaml")).eval().build_plan(1, 320, 320, "cpu", torch.float32).fusion_report()
    assert r32["stem"] == "staging + conv" and r32["dmff_blocks"] == "per-layer" and r32["detect"] == "conv, decode"


def test_chain_plans_for_host_fed_pipelines_have_no_graph_branches():
    """Model.plan_for(branches=False): the same launches in the same order, none of them on a parallel branch of the hipGraph, cached beside
    (not instead of) the branched plan of the same shape — DetectionPipeline(u8=True) builds its plans this way."""
    m = Model(load_cfg("yolov5s_Transfusion_kaist.yaml")).eval()
    m.compute_dtype = torch.bfloat16
    m.autotune = False
    a = m.plan_for(1, 320, 320, "cpu")
    b = m.plan_for(1, 320, 320, "cpu", branches=False)
    assert a is not b and a is m.plan_for(1, 320, 320, "cpu") and b is m.plan_for(1, 320, 320, "cpu", branches=False)
    assert a.branches and any(l.branch for l in a.launches)
    assert not b.branches and not any(l.branch for l in b.launches)
    assert [l.name for l in a.launches] == [l.name for l in b.launches]
