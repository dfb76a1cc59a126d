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
    u8 = names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16, u8=True))
    assert u8 == base                                                        # same launches from the uint8 batch
    try:
        Conv.fuse_stem2 = False
        assert names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))[:2] == ["stem", "conv3x3s2+1x1"]
        Conv.fuse_stem2, C3.fuse_cv3 = True, False
        assert names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))[1:3] == ["bottleneck", "conv1x1s1"]
        C3.fuse_cv3, C3.chain_bottlenecks = True, False
        assert "conv3x3s1+1x1" not in names(m.build_plan(2, 320, 320, "cpu", torch.bfloat16))
    finally:
        Conv.fuse_stem2, C3.fuse_cv3, C3.chain_bottlenecks = True, True, True
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
