"""Pin the CPU oracle (oracle/icaf_oracle.py) to the real reference through the committed fixtures.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py, which imports the reference
read-only and runs it on the CPU.  Weights / inputs are regenerated from icafusion_amd.synth.  CPU only.
"""
import numpy as np
import pytest
import torch

from helpers import load_cfg, load_golden, sample_idx
from icafusion_amd.models.yolo import Model
from icafusion_amd.synth import synth_images, synth_state_dict, synth_tensor
from oracle import icaf_oracle as oracle

MODEL_CASES = ["model_s_kaist_320_b2", "model_s_kaist_384x320_loops3", "model_l_vedai_320_b1",
               "model_s_kaist_640_b1", "model_s_add_kaist_320_b1", "model_n_ninfusion_flir_320_b2", "model_m_kaist_320_b1",
               "model_n_flir_352x320_b2",
               "model_s_kaist_544x672_b1"]     # the rect validation batch shape of KAIST (DMFF windows (11, 8) / (4, 12) / (8, 3))


@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_forward_matches_reference(name):
    g = load_golden(name)
    batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    cfg = load_cfg(str(g["yaml"]))
    ours = Model(cfg)                                   # CPU construction: parameter layout only
    sd = synth_state_dict(ours, seed)
    rgb, ir = synth_images(batch, h, w, seed)
    om = oracle.OracleModel(cfg, sd, loops=None if loops < 0 else loops)
    (z, logits, raws), outs = om.forward(rgb, ir, keep_layers=True)
    # every layer, sampled exactly where the fixture sampled the reference
    for i, o in enumerate(outs[:-1]):
        ref = g[f"layer{i}"]
        assert tuple(g[f"layer{i}_shape"]) == tuple(o.shape), f"layer {i} shape"
        got = o.reshape(-1)[torch.from_numpy(sample_idx(o.numel(), i))].numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 2e-4 * scale, f"layer {i}: {np.abs(got - ref).max()}"
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=2e-4, atol=2e-3)     # boxes are O(100 px)
    np.testing.assert_allclose(z.numpy()[..., 4:], g["z"][..., 4:], rtol=0, atol=1e-4)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=5e-4)
    for l, r in enumerate(raws):
        got = r.reshape(-1)[torch.from_numpy(sample_idx(r.numel(), 100 + l))].numpy()
        np.testing.assert_allclose(got, g[f"raw{l}"], rtol=1e-4, atol=5e-4)


DMFF_CASES = ["dmff_c128_20x20_in40x40", "dmff_c256_16x16_in40x40_overlap", "dmff_c128_20x20_in64x80_rect_loops3",
              "dmff_c512_10x10_in10x10_identity"]


def dmff_inputs(c, batch, h, w, seed):
    g = np.random.default_rng([seed, 77, c, h, w])
    rgb = torch.from_numpy(g.normal(0, 1, (batch, c, h, w)).astype(np.float32))
    ir = torch.from_numpy(g.normal(0, 1, (batch, c, h, w)).astype(np.float32))
    return rgb, ir


def dmff_state_dict(c, va, ha, seed):
    from icafusion_amd.models.common import TransformerFusionBlock
    blk = TransformerFusionBlock(c, va, ha)
    return {"model.20." + k: (v if k.endswith("num_batches_tracked") else synth_tensor("model.20." + k, v.shape, seed=seed))
            for k, v in blk.state_dict().items()}


@pytest.mark.parametrize("name", DMFF_CASES)
def test_dmff_block_matches_reference(name):
    g = load_golden(name)
    c, va, ha, batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    sd = dmff_state_dict(c, va, ha, seed)
    rgb, ir = dmff_inputs(c, batch, h, w, seed)
    pre = "model.20"
    with torch.no_grad():
        tv = oracle.pooled_tokens(rgb, va, ha, sd[pre + ".vis_coefficient.w1"], sd[pre + ".vis_coefficient.w2"],
                                  sd[pre + ".pos_emb_vis"])
        ti = oracle.pooled_tokens(ir, va, ha, sd[pre + ".ir_coefficient.w1"], sd[pre + ".ir_coefficient.w2"],
                                  sd[pre + ".pos_emb_ir"])
        av, ai = oracle.cross_attention(tv, ti, sd, pre + ".crosstransformer.0.crossatt", 8)
        ov, oi = oracle.cross_transformer(tv, ti, sd, pre + ".crosstransformer.0", 8, loops)
        out = oracle.dmff(rgb, ir, sd, pre, va, ha, 8, loops)
    for j, (k, t) in enumerate([("out", out), ("tok_in", tv), ("tok_in_ir", ti), ("tok_out", ov),
                                ("tok_out_ir", oi), ("att_v", av), ("att_i", ai)]):
        assert tuple(g[k + "_shape"]) == tuple(t.shape), k
        got = t.reshape(-1)[torch.from_numpy(sample_idx(t.numel(), 200 + j, 8192))].numpy()
        np.testing.assert_allclose(got, g[k], rtol=2e-4, atol=2e-4, err_msg=k)


def _kw(g):
    return dict(eval(str(g["kw"])))


@pytest.mark.parametrize("name,src", [("nms_s_conf25", "model_s_kaist_320_b2"),
                                      ("nms_s_conf30", "model_s_kaist_320_b2"),
                                      ("nms_s_conf001_iou5", "model_s_kaist_320_b2"),
                                      ("nms_l_multilabel", "model_l_vedai_320_b1"),
                                      ("nms_l_agnostic_classes", "model_l_vedai_320_b1")])
def test_nms_wrapper_matches_reference(name, src):
    """Wrapper logic (filtering, multi-label, class offsets, max_det) vs the reference's non_max_suppression run
    with the oracle's greedy core injected for torchvision.ops.nms — the core itself is parity-unpinned."""
    g, z = load_golden(name), load_golden(src)["z"]
    out = oracle.non_max_suppression(z, **_kw(g))
    assert len(out) == int(g["n"])
    for i, o in enumerate(out):
        np.testing.assert_array_equal(o, g[f"det{i}"])


def test_nms_core_c_and_numpy_agree():
    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 600, (3000, 2)).astype(np.float32)
    wh = rng.uniform(4, 200, (3000, 2)).astype(np.float32)
    boxes = np.concatenate((xy, xy + wh), 1)
    scores = rng.random(3000).astype(np.float32)
    scores[100:200] = scores[100]                 # ties: stable order by index
    a = oracle.nms_greedy(boxes, scores, 0.5)
    lib, oracle._NMS_LIB = oracle._NMS_LIB, False
    try:
        b = oracle.nms_greedy(boxes, scores, 0.5)
    finally:
        oracle._NMS_LIB = lib
    np.testing.assert_array_equal(a, b)
    assert len(a) > 10 and len(set(a.tolist())) == len(a)


def test_metrics_match_reference():
    g = load_golden("metrics_ap")
    ap, classes = oracle.ap_per_class(g["tp"], g["conf"], g["pcls"], g["tcls"])
    np.testing.assert_allclose(ap, g["ap"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(classes, g["classes"])
    np.testing.assert_allclose(oracle.box_iou(g["box1"], g["box2"]), g["iou"], rtol=1e-5, atol=1e-6)


def test_match_predictions_pinned_by_the_reference_block():
    """tests/golden/match_predictions.npz = outputs of the reference's own inline TP-matching block (test.py:196-230, exec'ed
    from the reference tree by make_golden.py): the oracle's statement and the host statement the device kernel is tested
    against must reproduce its flags exactly, from the same native-space boxes."""
    from icafusion_amd.utils.metrics import match_predictions as host_match
    from icafusion_amd.utils.general import scale_coords
    g = load_golden("match_predictions")
    iouv = g["iouv"]
    hits = 0
    for t in range(int(g["n"])):
        pred, predn, tbox, labels, want = g[f"pred{t}"], g[f"predn{t}"], g[f"tbox{t}"], g[f"labels{t}"], g[f"correct{t}"]
        H, W, h0, w0, gain, pw, ph = g[f"geom{t}"]
        mine = torch.from_numpy(pred.copy())
        scale_coords((int(H), int(W)), mine[:, :4], (int(h0), int(w0)), ((gain, gain), (pw, ph)))       # our scale_coords == the reference's
        np.testing.assert_array_equal(mine.numpy(), predn)
        gt = np.concatenate((labels[:, :1], tbox), 1) if len(labels) else np.zeros((0, 5), np.float32)
        np.testing.assert_array_equal(oracle.match_predictions(predn, gt, iouv), want, err_msg=f"oracle, trial {t}")
        np.testing.assert_array_equal(host_match(predn, gt, iouv), want, err_msg=f"host statement, trial {t}")
        hits += int(want.sum())
    assert hits > 100


def test_adaptive_pool_window_rule_over_a_sweep_of_sizes():
    """AdaptivePool2d (models/common.py:868-891) on ~3,800 (anchor grid, h, w) combinations recorded from the reference module:
    the host mirror's window(), the oracle's pooling and the reference agree on output shape and values — including the sizes where
    one side is above the grid and the other below it (the reference's stride becomes 0 and torch raises; the mirror raises too)
    and the identity case (both sides at or below the grid)."""
    import torch.nn.functional as F
    from icafusion_amd.models.common import AdaptivePool2d
    rows = load_golden("adaptive_pool_windows")["rows"]
    seen = {"pool": 0, "identity": 0, "raise": 0, "overlap": 0}
    for va, ha, h, w, oh, ow, s_avg, s_max in rows:
        va, ha, h, w, oh, ow = (int(v) for v in (va, ha, h, w, oh, ow))
        mirror = AdaptivePool2d(va, ha)
        if oh < 0:
            with pytest.raises(ValueError):
                mirror.window(h, w)
            seen["raise"] += 1
            continue
        th, tw, kh, kw, sh, sw = mirror.window(h, w)
        assert (th, tw) == (oh, ow), (va, ha, h, w)
        x = torch.from_numpy(np.random.default_rng([va, h, w]).normal(0, 1, (1, 2, h, w)).astype(np.float32))
        one, zero = torch.tensor(1.0), torch.tensor(0.0)
        pos = torch.zeros(1, oh * ow, 2)
        a = oracle.pooled_tokens(x, va, ha, one, zero, pos)               # avg only
        m = oracle.pooled_tokens(x, va, ha, zero, one, pos)               # max only
        assert a.shape == (1, oh * ow, 2)
        assert float(a.double().sum()) == pytest.approx(s_avg, rel=1e-6, abs=1e-4) and float(m.double().sum()) == pytest.approx(s_max, rel=1e-6, abs=1e-4)
        if h > va or w > ha:
            # the mirror's window describes the same pooling: F.avg_pool2d with its kernel / stride gives the oracle's tokens
            ref = F.avg_pool2d(x, (kh, kw), (sh, sw)).reshape(1, 2, -1).permute(0, 2, 1)
            assert torch.equal(ref, a)
            seen["pool"] += 1
            seen["overlap"] += kh > sh or kw > sw
        else:
            assert (kh, kw, sh, sw) == (1, 1, 1, 1) and (th, tw) == (h, w)
            seen["identity"] += 1
    assert seen["pool"] > 3000 and seen["identity"] >= 15 and seen["raise"] > 100 and seen["overlap"] > 2000, seen


@pytest.mark.parametrize("tag", ["s", "l"])
def test_sixteen_bit_oracle_deviates_like_the_reference_in_sixteen_bit(tag):
    """The yardstick of the 16-bit parity bounds (tests/test_gpu_parity16.py, profiles/parity_16bit.json) is the oracle evaluated in
    bf16 / fp16 — `OracleModel(dtype=)`.  tests/golden/reference_16bit.npz holds the REAL reference, fused and cast with `.to(dtype)` as
    detect_twostream.py:33-40 does, run on the CPU: the oracle's deviation from fp32 must be the reference's (mean within 20 %; the
    maximum is a single element and varies with the summation order — the reference's own run moves by 10 % with the thread count)."""
    g = load_golden("reference_16bit")
    batch, h, w, seed = [int(v) for v in g[f"{tag}_meta"]]
    cfg = load_cfg(str(g[f"{tag}_yaml"]))
    m = Model(cfg).eval()
    m.load_state_dict(synth_state_dict(m, seed))
    fsd = m.fuse().state_dict()
    rgb, ir = synth_images(batch, h, w, seed)
    z32 = oracle.OracleModel(cfg, fsd).forward(rgb, ir)[0]
    ref32 = torch.from_numpy(g[f"{tag}_z32"])
    assert float((z32 - ref32).abs().max()) <= 2e-3                       # fp32: the oracle IS the reference (boxes are O(100 px))
    for dn, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        o16 = oracle.OracleModel(cfg, fsd, dtype=dt).forward(rgb, ir)[0].float()
        d = (o16 - z32).abs()
        mine = np.array([float(d[..., :4].max()), float(d[..., :4].mean()), float(d[..., 4:].max()), float(d[..., 4:].mean())])
        ratio = mine / g[f"{tag}_{dn}_dev"]
        assert 0.8 <= ratio[1] <= 1.25 and 0.8 <= ratio[3] <= 1.25, (tag, dn, ratio)          # mean box / conf deviation
        assert 0.4 <= ratio[0] <= 2.5 and 0.4 <= ratio[2] <= 2.5, (tag, dn, ratio)            # max box / conf deviation
        # and the two 16-bit evaluations are closer to each other than either is to fp32 (same roundings, different summation order)
        x = (o16 - torch.from_numpy(g[f"{tag}_{dn}_z16"])).abs()
        assert float(x[..., :4].mean()) < g[f"{tag}_{dn}_dev"][1] and float(x[..., 4:].mean()) < g[f"{tag}_{dn}_dev"][3]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_planted_detector_recipe_is_separated(dtype):
    """The weight recipe behind the 16-bit mAP test (tools/parity16.py: planted objects + a fitted objectness read-out) must itself
    be insensitive to 16-bit arithmetic: the REFERENCE evaluated in bf16 / fp16 (the oracle by torch in that type) keeps mAP@50
    within 0.1 of its fp32 value, at a non-trivial mAP.  (The GPU test asserts the same bound for the HIP path on 16 images.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity16
    rec = parity16.measure_map(dtype, B=4, hip=False)
    assert rec["objects"] >= 80 and rec["detections_oracle"] > 2 * rec["objects"]
    assert 50.0 < rec["map50_oracle_fp32"] < 95.0
    for st in rec["fit"][1:]:                              # P4 / P5: a clear gap between planted cells and background
        assert st["planted_cells_min"] - st["background_max"] > 0.5
    assert abs(rec["map50_delta_reference16"]) <= 0.1, rec


def test_small_object_recipe_depends_on_box_accuracy():
    """RECIPES["p3_small"] of tools/parity16.py (10 x 10 pixel objects at the P3 level): separated scores at a non-trivial mAP, and a metric
    that DOES move with localisation error — the reference evaluated in bf16 (boxes decoded into a bf16 tensor) keeps mAP@50 but loses more
    than two points of mAP@.5:.95.  The GPU test holds the HIP path to this recipe (tests/test_gpu_parity16.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity16
    rec = parity16.measure_map("bf16", B=4, hip=False, recipe="p3_small")
    box = rec["object_box_px"]
    assert rec["objects"] >= 300 and 8.0 <= box["w_median"] <= 16.0 and 8.0 <= box["h_median"] <= 16.0 and box["max"] <= 16.0
    assert 50.0 < rec["map50_oracle_fp32"] < 95.0
    assert rec["fit"][0]["planted_cells_min"] - rec["fit"][0]["background_max"] > 0.5
    assert abs(rec["map50_delta_reference16"]) <= 0.1 and rec["map_delta_reference16"] < -2.0, rec
