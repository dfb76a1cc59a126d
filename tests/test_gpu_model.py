"""End-to-end parity on the MI355X: Model(cfg) built from the repo's yaml files, weights from icafusion_amd.synth,
HIP forward vs (a) the committed outputs of the real reference (tests/golden) and (b) the CPU oracle at other sizes."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_cfg, load_golden, sample_idx          # noqa: E402
from icafusion_amd.models.common import C3, SPPF, Conv, TransformerFusionBlock  # noqa: E402
from icafusion_amd.models.yolo import Model                     # noqa: E402
from icafusion_amd.synth import synth_images, synth_labels, synth_state_dict, synth_tensor  # noqa: E402
from oracle import icaf_oracle as oracle                        # noqa: E402

DEV = "cuda:0"


def build(yaml_name, seed, dtype=torch.float32, loops=None):
    cfg = load_cfg(yaml_name)
    m = Model(cfg).eval()
    sd = synth_state_dict(m, seed)
    m.load_state_dict(sd)
    if loops is not None:
        for i in (20, 21, 22):
            m.model[i].crosstransformer[0].loops = loops
    m = m.to(DEV)
    if dtype != torch.float32:
        m.compute_dtype = dtype          # fp32 masters, 16-bit packed weights / activations
    return cfg, sd, m


GOLDEN = ["model_s_kaist_320_b2", "model_s_kaist_384x320_loops3", "model_l_vedai_320_b1", "model_s_kaist_640_b1",
          "model_s_add_kaist_320_b1", "model_n_ninfusion_flir_320_b2",      # Add / NiNfusion variants
          "model_m_kaist_320_b1",                                           # yolov5m widths: 48 / 96 / 192 / 384 / 768 channels
          "model_n_flir_352x320_b2",                                        # yolov5n + DMFF, rectangular input, FLIR classes
          "model_s_kaist_544x672_b1"]                                       # test.py's rect validation batch shape of KAIST frames: DMFF windows
#                                                                             (11, 8) / (4, 12) / (8, 3), odd 17 x 21 map at P5 (utils/datasets.py:840-849)

# Measured fp32 errors of every golden (absolute: box pixels, scores, logits, raw maps) are appended to gpurun_out/parity_fp32.jsonl;
# profiles/parity_fp32.json is the committed copy of one run.  Bound = 10 x the committed measurement (never below the floor, which
# is what a different summation order of one more layer would cost), absolute - not relative to max|z|: a regression of a few
# hundredths of a pixel fails.  A golden without a committed measurement falls back to north_star's 1e-3 relative to the scale.
_FLOOR = {"box_px": 2e-3, "score": 2e-5, "logit": 2e-4, "raw": 2e-4}


def _fp32_bounds(name, scale):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "parity_fp32.json")
    try:
        with open(path) as f:
            rec = {r["golden"]: r for r in json.load(f)["goldens"]}[name]
        return {k: max(10.0 * rec[k], _FLOOR[k]) for k in _FLOOR}
    except (OSError, KeyError):
        return {"box_px": 1e-3 * scale["box_px"], "score": 1e-3, "logit": 1e-3 * scale["logit"], "raw": 1e-3 * scale["raw"]}


def _assert_abs(z, ref, what, box_px=2e-2, score=2e-4):
    """Absolute fp32 bounds (pixels / probabilities): ~10 x what the goldens measure (profiles/parity_fp32.json)."""
    e_box, e_sc = float(np.abs(z[..., :4] - ref[..., :4]).max()), float(np.abs(z[..., 4:] - ref[..., 4:]).max())
    print(f"{what}: box error {e_box:.3g} px, score error {e_sc:.3g}")
    assert e_box <= box_px and e_sc <= score, (what, e_box, e_sc)


@pytest.mark.parametrize("name", GOLDEN)
def test_fp32_forward_matches_reference_golden(name):
    """fp32 HIP path vs the real reference's recorded output (north_star: fp32 1e-3; asserted far tighter, see _fp32_bounds)."""
    g = load_golden(name)
    batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    cfg, sd, m = build(str(g["yaml"]), seed, loops=None if loops < 0 else loops)
    rgb, ir = synth_images(batch, h, w, seed)
    z, logits, raws = m(rgb.to(DEV), ir.to(DEV))
    zc, ref = z.cpu().numpy(), g["z"]
    assert zc.shape == ref.shape
    err = {"box_px": float(np.abs(zc[..., :4] - ref[..., :4]).max()), "score": float(np.abs(zc[..., 4:] - ref[..., 4:]).max()),
           "logit": float(np.abs(logits.cpu().numpy() - g["logits"]).max()), "raw": 0.0}
    scale = {"box_px": max(1.0, float(np.abs(ref[..., :4]).max())), "logit": max(1.0, float(np.abs(g["logits"]).max())), "raw": 1.0}
    for l, r in enumerate(raws):
        assert tuple(r.shape) == tuple(g[f"raw{l}_shape"])
        got = r.cpu().reshape(-1)[torch.from_numpy(sample_idx(r.numel(), 100 + l))].numpy()
        err["raw"] = max(err["raw"], float(np.abs(got - g[f"raw{l}"]).max()))
        scale["raw"] = max(scale["raw"], float(np.abs(g[f"raw{l}"]).max()))
    bound = _fp32_bounds(name, scale)
    rec = dict(err, golden=name, yaml=str(g["yaml"]), batch=batch, height=h, width=w, box_px_scale=scale["box_px"], bound=bound)
    print(json.dumps(rec))
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_fp32.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    for k in err:
        assert err[k] <= bound[k], f"{name}: {k} error {err[k]:.3g} > {bound[k]:.3g}"


def test_loops_yaml_argument_equals_attribute():
    """The optional 4th DMFF yaml argument (loops) gives the same result as setting .loops on the reference class."""
    g = load_golden("model_s_kaist_384x320_loops3")
    batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    cfg, sd, m = build("yolov5s_Transfusion_kaist_loops3.yaml", seed)
    assert m.model[20].crosstransformer[0].loops == 3
    rgb, ir = synth_images(batch, h, w, seed)
    z = m(rgb.to(DEV), ir.to(DEV))[0].cpu().numpy()
    _assert_abs(z, g["z"], "loops yaml")


def test_fp32_forward_matches_oracle_other_shape():
    """A shape with no golden file (512x640, batch 3): HIP vs oracle, both fed identical weights."""
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", seed=11)
    rgb, ir = synth_images(3, 512, 640, seed=11)
    ref = oracle.OracleModel(cfg, sd).forward(rgb, ir)[0].numpy()
    z = m(rgb.to(DEV), ir.to(DEV))[0].cpu().numpy()
    _assert_abs(z, ref, "512x640 batch 3 vs oracle")


def test_fused_model_same_output():
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", seed=4)
    rgb, ir = synth_images(1, 320, 320, seed=4)
    a = m(rgb.to(DEV), ir.to(DEV))[0]
    b = m.fuse()(rgb.to(DEV), ir.to(DEV))[0]
    assert not any(hasattr(c, "bn") for c in m.modules() if type(c) is Conv)
    assert float((a - b).abs().max()) <= 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_low_precision_forward_tolerance(dtype):
    """bf16 / f16 throughput builds vs the reference's fp32 output (golden), bounded by the reference's OWN deviation in the same
    16-bit type on the same weights and inputs (the oracle evaluated by torch in that type = `model.half()`): at most 1.5x its
    maximum and mean errors — see tests/test_gpu_parity16.py for the full-size configurations and the rationale."""
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", seed=1, dtype=dtype)
    rgb, ir = synth_images(2, 320, 320, seed=1)
    ref = load_golden("model_s_kaist_320_b2")["z"]
    ref16 = oracle.OracleModel(cfg, sd, dtype=dtype).forward(rgb, ir)[0].float().numpy()
    z = m(rgb.to(DEV), ir.to(DEV))[0].float().cpu().numpy()
    assert np.isfinite(z).all()
    for name, sl in (("box px", np.s_[..., :4]), ("score", np.s_[..., 4:])):
        e_hip, e_ref = np.abs(z[sl] - ref[sl]), np.abs(ref16[sl] - ref[sl])
        print(f"{dtype} {name}: HIP max {e_hip.max():.4g} mean {e_hip.mean():.4g} | reference-in-{dtype} max {e_ref.max():.4g} mean {e_ref.mean():.4g}")
        assert e_hip.max() <= 1.5 * e_ref.max() and e_hip.mean() <= 1.5 * e_ref.mean()


def test_forward_u8_equals_float_forward():
    """Model.forward_u8 (uint8 6-channel batch) == forward(img[:, :3] / 255, img[:, 3:] / 255)."""
    cfg = load_cfg("yolov5s_Transfusion_kaist.yaml")
    model = Model(cfg).eval()
    model.load_state_dict(synth_state_dict(model, seed=3))
    model = model.to("cuda:0")
    g = np.random.default_rng(5)
    img6 = torch.from_numpy(g.integers(0, 256, (2, 6, 320, 384), dtype=np.uint8)).cuda()
    f = (img6.cpu().float() / 255.0).cuda()      # true division, as the reference's CPU path
    for paired in (True, False):
        model.pair_streams = paired
        model.invalidate()
        z0 = model(f[:, :3].contiguous(), f[:, 3:].contiguous())[0]
        z1 = model.forward_u8(img6)[0]
        assert torch.equal(z0, z1)


def test_paired_streams_equal_separate_streams():
    """Running both backbones as groups=2 launches (and C3's fused cv1+cv2 GEMM) must not change a single bit."""
    cfg = load_cfg("yolov5s_Transfusion_kaist.yaml")
    model = Model(cfg).eval()
    model.load_state_dict(synth_state_dict(model, seed=4))
    model = model.to("cuda:0")
    rgb, ir = synth_images(2, 320, 320, seed=4)
    outs = []
    for paired in (True, False):
        model.pair_streams = paired
        model.invalidate()
        outs.append(model(rgb.cuda(), ir.cuda())[0])
    assert torch.equal(outs[0], outs[1])


def test_fused_dmff_tail_matches_materialised_tail():
    """conv1x1_out(cat(f + up(t))) computed as conv(cat(f)) + up(conv(cat(t))) (one GEMM, no merged tensor) vs the
    reference's order of operations (upsample_merge kernel + GEMM): equal up to fp32 rounding."""
    cfg = load_cfg("yolov5s_Transfusion_kaist.yaml")
    model = Model(cfg).eval()
    model.load_state_dict(synth_state_dict(model, seed=6))
    model = model.to("cuda:0")
    rgb, ir = synth_images(2, 384, 320, seed=6)
    outs = []
    for fused in (True, False):
        for i in (20, 21, 22):
            model.model[i].fuse_tail = fused
        model.invalidate()
        plan = model.plan_for(2, 384, 320)
        names = [l.name for l in plan.launches]
        assert ("dmff_tail_tokens" in names) == fused and ("dmff_upsample_merge" in names) == (not fused)
        outs.append(model(rgb.cuda(), ir.cuda())[0])
    err = (outs[0] - outs[1]).abs().max().item() / outs[1].abs().max().item()
    assert err < 1e-5, err


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_bottleneck_model_matches_two_launch_model(dtype):
    """Fused Bottlenecks (one launch, 3-slot C3 buffer, swapped cv3 columns) vs the two-launch form: the Bottlenecks are
    bit-identical, cv3 then sums its K in a different order, so outputs agree to 16-bit rounding noise."""
    from icafusion_amd.models.common import Bottleneck
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", 7, dtype)
    rgb, ir = synth_images(2, 640, 640, seed=7)
    outs = []
    Bottleneck.fuse_widths = (32, 64)            # exercise every built width, not only the one the plan uses by default
    for fuse in (True, False):
        Bottleneck.fuse = fuse
        m.invalidate()
        names = [l.name for l in m.plan_for(2, 640, 640).launches]
        assert ("bottleneck" in names) == fuse
        outs.append(m(rgb.cuda(), ir.cuda())[0].float())
    Bottleneck.fuse, Bottleneck.fuse_widths = True, (32,)
    scale = outs[1][..., :4].abs().max().item()
    assert (outs[0][..., :4] - outs[1][..., :4]).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 3e-3) * scale
    assert (outs[0][..., 4:] - outs[1][..., 4:]).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 3e-3)


def test_reference_style_half_model_and_half_inputs():
    """`model.half()` + `img.half()` exactly as detect_twostream.py:33-40,73-80 / test.py:73-75,116-118 drive the model."""
    cfg, sd, m32 = build("yolov5s_Transfusion_kaist.yaml", 9)
    rgb, ir = synth_images(1, 320, 320, seed=9)
    z32 = m32(rgb.cuda(), ir.cuda())[0]
    cfg, sd, mh = build("yolov5s_Transfusion_kaist.yaml", 9)
    mh = mh.half()
    assert next(mh.parameters()).dtype == torch.float16 and mh.model[-1].anchor_grid.dtype == torch.float32
    zh = mh(rgb.cuda().half(), ir.cuda().half())[0]
    assert zh.dtype == torch.float32 and torch.isfinite(zh).all()
    assert (zh[..., :4] - z32[..., :4]).abs().max().item() <= 2.0            # pixels (fp16 weights + activations)
    assert (zh[..., 4:] - z32[..., 4:]).abs().max().item() <= 1e-2


@pytest.mark.parametrize("shape", [(1, 320, 320), (3, 352, 416), (2, 640, 512), (5, 384, 640)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_all_fusions_on_vs_off_across_shapes(shape, dtype):
    """Every plan-level optimisation (paired streams, persistent stem, fused Bottleneck, chained Conv -> C3, fused DMFF
    tail, graph branches + hipGraph replay) against the plain launch-per-layer plan, on batch sizes and image sizes that
    leave ragged patches / tiles everywhere.  16-bit results agree to rounding noise (cv3 sums K in another order)."""
    from icafusion_amd.models.common import Bottleneck
    B, H, W = shape
    cfg, sd, m = build("yolov5s_Transfusion_FLIR.yaml", 13, dtype)
    rgb, ir = synth_images(B, H, W, seed=13)
    outs = []
    for on in (True, False):
        Conv.fuse_stem = Conv.chain_fuse = Bottleneck.fuse = on
        m.pair_streams = m.branch_dmff = m.use_graph = on
        for i in (20, 21, 22):
            m.model[i].fuse_tail = on
        m.invalidate()
        names = [l.name for l in m.plan_for(B, H, W).launches]
        assert any(n.startswith("stem") for n in names) == on and (any(n.endswith("+1x1") for n in names)) == on
        outs.append(m(rgb.cuda(), ir.cuda())[0].float())
    Conv.fuse_stem = Conv.chain_fuse = Bottleneck.fuse = True
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
    scale = outs[1][..., :4].abs().max().item()
    assert torch.isfinite(outs[0]).all()
    assert (outs[0][..., :4] - outs[1][..., :4]).abs().max().item() <= tol * scale
    assert (outs[0][..., 4:] - outs[1][..., 4:]).abs().max().item() <= tol


@pytest.mark.parametrize("shape", [(2, 320, 320), (3, 352, 416), (1, 640, 512)])
@pytest.mark.parametrize("u8", [False, True])
def test_stem2_plan_is_bit_identical_to_three_launch_plan(shape, u8):
    """Rows 0-2a as one persistent kernel (Conv.fuse_stem2) vs stem -> conv3x3/s2 with chained cv1 | cv2: the whole
    network output is bit-identical (the intermediate tensors are rounded at the same points), fp32 and uint8 inputs."""
    B, H, W = shape
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", 17, torch.bfloat16)
    rgb, ir = synth_images(B, H, W, seed=17)
    img6 = (torch.cat((rgb, ir), 1) * 255).round().to(torch.uint8).cuda()
    outs = []
    try:
        for on in (True, False):
            Conv.fuse_stem2 = on
            m.invalidate()
            names = [l.name for l in m.plan_for(B, H, W, u8=u8).launches]
            assert (names[0] == "stem+conv3x3s2+1x1") == on
            outs.append((m.forward_u8(img6) if u8 else m(rgb.cuda(), ir.cuda()))[0].clone())
    finally:
        Conv.fuse_stem2 = True
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("shape", [(2, 320, 320), (3, 352, 416)])
def test_bottleneck_cv3_plan_is_bit_identical(shape):
    """C3 of row 2 with Bottleneck + cv3 as one launch (C3.fuse_cv3) vs Bottleneck launch + cv3 launch: identical output."""
    from icafusion_amd.models.common import C3
    B, H, W = shape
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", 19, torch.bfloat16)
    rgb, ir = synth_images(B, H, W, seed=19)
    outs = []
    try:
        for on in (True, False):
            C3.fuse_cv3 = on
            m.invalidate()
            names = [l.name for l in m.plan_for(B, H, W).launches]
            assert ("bottleneck+cv3" in names) == on
            outs.append(m(rgb.cuda(), ir.cuda())[0].clone())
    finally:
        C3.fuse_cv3 = True
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["yolov5s_Transfusion_kaist.yaml", "yolov5l_Transfusion_kaist.yaml"])
def test_chained_bottlenecks_plan_is_bit_identical(name):
    """C3 blocks with n > 1: each Bottleneck's 3x3 (+ shortcut) launch also computes the next Bottleneck's 1x1
    (C3.chain_bottlenecks) vs separate launches: identical output."""
    from icafusion_amd.models.common import C3
    cfg, sd, m = build(name, 23, torch.bfloat16)
    rgb, ir = synth_images(2, 320, 352, seed=23)
    outs, counts = [], []
    try:
        for on in (True, False):
            C3.chain_bottlenecks = on
            m.invalidate()
            counts.append(len(m.plan_for(2, 320, 352).launches))
            outs.append(m(rgb.cuda(), ir.cuda())[0].clone())
    finally:
        C3.chain_bottlenecks = True
    assert counts[0] < counts[1]
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name,dtype,shape", [("yolov5s_Transfusion_kaist.yaml", torch.bfloat16, (2, 320, 352)),
                                              ("yolov5s_Transfusion_kaist.yaml", torch.float16, (3, 352, 416)),
                                              ("yolov5l_Transfusion_kaist.yaml", torch.bfloat16, (2, 320, 320))])
def test_c3_tail_plan_is_bit_identical(name, dtype, shape):
    """C3 blocks whose last Bottleneck is a 128 -> 128 3x3: cv3 rides on that launch (Conv.chain_tail; icaf_conv_args.x2) vs the
    separate cv3 GEMM over cat(m, cv2): one launch less per such C3, identical network output (backbone blocks with the shortcut
    and both streams in one launch, head blocks without)."""
    B, H, W = shape
    cfg, sd, m = build(name, 29, dtype)
    rgb, ir = synth_images(B, H, W, seed=29)
    outs, tails, counts = [], [], []
    try:
        for on in (True, False):
            Conv.chain_tail = on
            m.invalidate()
            names = [l.name for l in m.plan_for(B, H, W).launches]
            tails.append(names.count("conv3x3s1+cv3"))
            counts.append(len(names))
            outs.append(m(rgb.cuda(), ir.cuda())[0].clone())
    finally:
        Conv.chain_tail = True
    assert tails[0] >= 2 and tails[1] == 0 and counts[1] - counts[0] == tails[0]
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_folded_upsample_matches_materialised_concat(dtype):
    """Head rows Upsample -> Concat -> C3 with the up-sampled half of the C3's 1x1 computed at low resolution
    (Model.fold_upsample) vs the materialised concat: fp32 agrees to summation-order noise, bf16 to rounding noise."""
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", 29, dtype)
    rgb, ir = synth_images(2, 320, 384, seed=29)
    outs = []
    for on in (True, False):
        m.fold_upsample = on
        m.invalidate()
        names = [l.name for l in m.plan_for(2, 320, 384).launches]
        assert ("c3_up_term" in names) == on and ("upsample_nearest" in names) != on
        outs.append(m(rgb.cuda(), ir.cuda())[0].float())
    m.fold_upsample = False
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    scale = outs[1][..., :4].abs().max().item()
    assert (outs[0][..., :4] - outs[1][..., :4]).abs().max().item() <= tol * scale
    assert (outs[0][..., 4:] - outs[1][..., 4:]).abs().max().item() <= tol


def test_graph_replay_equals_eager():
    cfg, sd, m = build("yolov5s_Transfusion_kaist.yaml", seed=1, dtype=torch.bfloat16)
    rgb, ir = synth_images(2, 320, 320, seed=1)
    a = m(rgb.to(DEV), ir.to(DEV))[0]
    m.invalidate()
    m.use_graph = True
    b = m(rgb.to(DEV), ir.to(DEV))[0]
    c = m(rgb.to(DEV), ir.to(DEV))[0]
    assert torch.equal(a, b) and torch.equal(b, c)


def test_modules_standalone_duck_typing():
    """Each module is still a callable nn.Module on NCHW tensors (the reference's boundary)."""
    g = np.random.default_rng(5)
    x = torch.from_numpy(g.normal(0, 1, (2, 64, 24, 20)).astype(np.float32))

    def filled(mod, prefix, seed=9):
        sd = {k: (v if k.endswith("num_batches_tracked") else synth_tensor(prefix + k, v.shape, seed=seed))
              for k, v in mod.state_dict().items()}
        mod.load_state_dict(sd)
        for b in mod.modules():
            if isinstance(b, torch.nn.BatchNorm2d):
                b.eps = 1e-3
        return {prefix + k: v for k, v in sd.items()}, mod.eval().to(DEV)

    sd, conv = filled(Conv(64, 96, 3, 2), "model.1.")
    got = conv(x.to(DEV)).cpu()
    ref = oracle.conv_bn_silu(x, sd, "model.1", 3, 2, 1)
    assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1e-4
    sd, c3 = filled(C3(64, 64, 2), "model.2.")
    ref = oracle.c3(x, sd, "model.2", 2, True)
    assert float((c3(x.to(DEV)).cpu() - ref).abs().max()) <= 1e-4
    sd, sp = filled(SPPF(64, 64, 5), "model.9.")
    ref = oracle.sppf(x, sd, "model.9", 5)
    assert float((sp(x.to(DEV)).cpu() - ref).abs().max()) <= 1e-4
    with pytest.raises(RuntimeError):
        conv(x)                                   # CPU tensor: no silent fallback
    with pytest.raises(NotImplementedError):
        conv.train()(x.to(DEV))


@pytest.mark.parametrize("name", ["dmff_c128_20x20_in40x40", "dmff_c256_16x16_in40x40_overlap",
                                  "dmff_c128_20x20_in64x80_rect_loops3", "dmff_c512_10x10_in10x10_identity"])
def test_dmff_block_matches_reference_golden(name):
    g = load_golden(name)
    c, va, ha, batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    blk = TransformerFusionBlock(c, va, ha, loops_num=loops)
    sd = {k: (v if k.endswith("num_batches_tracked") else synth_tensor("model.20." + k, v.shape, seed=seed))
          for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk.conv1x1_out.bn.eps = 1e-3
    blk = blk.eval().to(DEV)
    rg = np.random.default_rng([seed, 77, c, h, w])
    rgb = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32))
    ir = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32))
    out = blk([rgb.to(DEV), ir.to(DEV)]).cpu()
    assert tuple(out.shape) == tuple(g["out_shape"])
    got = out.reshape(-1)[torch.from_numpy(sample_idx(out.numel(), 200, 8192))].numpy()
    assert np.abs(got - g["out"]).max() <= 1e-3 * max(1.0, np.abs(g["out"]).max())


def test_map50_matches_oracle_on_synthetic_labels():
    """mAP@0.5 of HIP detections vs oracle detections against the same synthetic labels (north_star: within 0.1)."""
    from icafusion_amd.utils.general import non_max_suppression
    cfg, sd, m = build("yolov5s_Transfusion_FLIR.yaml", seed=6)
    B, H, W = 4, 320, 320
    rgb, ir = synth_images(B, H, W, seed=6)
    labels = synth_labels(B, 3, seed=6).numpy()
    zr = oracle.OracleModel(cfg, sd).forward(rgb, ir)[0].numpy()
    zg = m(rgb.to(DEV), ir.to(DEV))[0]
    dets_g = [d.cpu().numpy() for d in non_max_suppression(zg, 0.001, 0.5, multi_label=True)]
    dets_r = oracle.non_max_suppression(zr, 0.001, 0.5, multi_label=True)
    iouv = np.linspace(0.5, 0.95, 10)

    def map50(dets):
        tp, conf, pcls, tcls = [], [], [], []
        for b in range(B):
            gt = labels[labels[:, 0] == b][:, 1:].copy()
            box = gt[:, 1:5] * np.array([W, H, W, H], np.float32)
            gt[:, 1:5] = np.concatenate((box[:, :2] - box[:, 2:] / 2, box[:, :2] + box[:, 2:] / 2), 1)
            tp.append(oracle.match_predictions(dets[b], gt, iouv)); conf.append(dets[b][:, 4]); pcls.append(dets[b][:, 5])
            tcls.append(gt[:, 0])
        ap, _ = oracle.ap_per_class(np.concatenate(tp), np.concatenate(conf), np.concatenate(pcls), np.concatenate(tcls))
        return 100.0 * ap[:, 0].mean()

    a, b = map50(dets_g), map50(dets_r)
    assert abs(a - b) <= 0.1, f"mAP@50 {a:.3f} vs {b:.3f}"
