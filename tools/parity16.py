#!/usr/bin/env python3
"""16-bit parity of the HIP path on the BENCHMARKED configurations (BASELINE.json configs 2-5, per-GPU shard sizes).

For each configuration the full-size batch runs on the MI355X in the configuration's 16-bit type; a slice of it is compared
with the fp32 CPU oracle (= the reference's algorithm, pinned by tests/golden) on the same weights and inputs:

    err_hip  = | z_hip16[slice]  - z_oracle_fp32 |        box coordinates in pixels, scores (obj, cls) absolute
    err_ref  = | z_oracle16      - z_oracle_fp32 |        the REFERENCE's own 16-bit deviation: the oracle evaluated by torch
                                                          in the same 16-bit type, as `model.half()` does (detect_twostream.py:40,
                                                          test.py:73-75; oracle/icaf_oracle.py OracleModel(dtype=))

The tolerance for err_hip is a stated multiple of err_ref measured on the SAME weights and inputs (tests/test_gpu_parity16.py).
BASELINE.md quotes err_ref for the reference's default initialisation (bf16 2.3 px / 1.9e-3, fp16 0.25 px / 2.3e-4 at 640x640,
yolov5s); the synthetic weights of icafusion_amd.synth spread the Detect logits much wider (their purpose: non-degenerate
detections), which raises both errors alike — hence the yardstick is re-measured rather than quoted.

Also: mAP@50 of the 16-bit HIP detections vs the fp32 oracle's detections through the same ap_per_class on a detector with
SEPARATED scores (planted objects + a fitted objectness read-out, see planted_detector below), next to the same number for the
reference evaluated in the 16-bit type: north_star's "mAP@50 within 0.1" is asserted without a yardstick multiplier.

    python tools/parity16.py [--out gpurun_out/parity_16bit.json] [--only c2,c4]      # on the GPU box

Test infrastructure: imports oracle/ as the checker.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402
import yaml          # noqa: E402

DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
# name -> (yaml, dtype, per-GPU batch, H, W, DMFF iterations, images of the batch compared with the oracle, seed)
CONFIGS = {
    "c2_s_bf16_b32_640": ("yolov5s_Transfusion_kaist.yaml", "bf16", 32, 640, 640, 1, (0, 31), 2),
    "c3_l_bf16_b32_640_shard": ("yolov5l_Transfusion_kaist.yaml", "bf16", 32, 640, 640, 1, (5, 30), 3),
    "c4_s_bf16_b64_512x640_loops3": ("yolov5s_Transfusion_kaist.yaml", "bf16", 64, 512, 640, 3, (0, 63), 4),
    "c5_l_vedai_f16_b16_1280_shard": ("yolov5l_Transfusion_VEDAI.yaml", "f16", 16, 1280, 1280, 1, (7,), 5),
    # the batch shape real KAIST frames take under test.py's rect protocol (utils/datasets.py:840-849): DMFF windows (11, 8) / (4, 12) /
    # (8, 3), odd 17 x 21 map at P5
    "kaist_rect_s_bf16_b16_544x672": ("yolov5s_Transfusion_kaist.yaml", "bf16", 16, 544, 672, 1, (0, 15), 7),
}


def load_cfg(name):
    with open(os.path.join(ROOT, "models", "transformer", name)) as f:
        return yaml.safe_load(f)


def build(yaml_name, dtype, loops, seed, dev="cuda:0"):
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_state_dict
    cfg = load_cfg(yaml_name)
    m = Model(cfg).eval()
    sd = synth_state_dict(m, seed)
    m.load_state_dict(sd)
    m.fuse()                                             # what attempt_load serves (models/experimental.py:119)
    fsd = {k: v.clone() for k, v in m.state_dict().items()}
    for blk in m.model:
        if hasattr(blk, "crosstransformer"):
            blk.crosstransformer[0].loops = loops
    m = m.to(dev)
    m.compute_dtype = DT[dtype]
    return cfg, fsd, m


def err_stats(z, ref):
    d = (z.double() - ref.double()).abs()
    box, sc = d[..., :4], d[..., 4:]
    return {"box_px_max": float(box.max()), "box_px_mean": float(box.mean()), "box_px_p999": float(np.quantile(box.flatten().numpy(), 0.999)),
            "score_max": float(sc.max()), "score_mean": float(sc.mean())}


def _cpu_threads():
    """torch's CPU kernels oversubscribe badly on these layers with every hardware thread of a 256-thread host (the same
    observation as bench.py's cpu_baseline): cap the oracle at 32."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def measure(name, with_reference16=True):
    """Run one configuration; returns the record written to profiles/parity_16bit.json.  with_reference16=False skips the
    reference-in-16-bit evaluation on the CPU (yolov5l at 1280x1280 in fp16 takes minutes on the host: the GPU test takes that
    yardstick from the committed profiles/parity_16bit.json instead)."""
    from icafusion_amd.synth import synth_images
    from oracle import icaf_oracle as oracle
    _cpu_threads()
    yaml_name, dtype, B, H, W, loops, pick, seed = CONFIGS[name]
    cfg, fsd, m = build(yaml_name, dtype, loops, seed)
    rgb, ir = synth_images(B, H, W, seed=seed)
    t0 = time.perf_counter()
    z = m(rgb.cuda(), ir.cuda())[0]
    torch.cuda.synchronize()
    assert torch.isfinite(z).all()
    idx = torch.tensor(pick)
    r, i = rgb[idx].contiguous(), ir[idx].contiguous()
    ref32 = oracle.OracleModel(cfg, fsd, loops=loops).forward(r, i)[0]
    hip = z[idx.cuda()].float().cpu()
    rec = {"config": name, "yaml": yaml_name, "dtype": dtype, "batch": B, "height": H, "width": W, "dmff_loops": loops,
           "images_compared": list(pick), "rows_per_image": int(z.shape[1]), "hip16_vs_oracle_fp32": err_stats(hip, ref32)}
    if with_reference16:
        ref16 = oracle.OracleModel(cfg, fsd, loops=loops, dtype=DT[dtype]).forward(r, i)[0].float()
        rec["reference16_vs_oracle_fp32"] = err_stats(ref16, ref32)
        a, b = rec["hip16_vs_oracle_fp32"], rec["reference16_vs_oracle_fp32"]
        rec["ratio_hip_over_reference16"] = {k: round(a[k] / max(b[k], 1e-12), 3) for k in a}
    rec["seconds"] = round(time.perf_counter() - t0, 1)
    del m
    torch.cuda.empty_cache()
    return rec


def _xyxy(lab, W, H):
    box = lab[:, 1:5] * np.array([W, H, W, H], np.float32)
    return np.concatenate((lab[:, :1], box[:, :2] - box[:, 2:] / 2, box[:, :2] + box[:, 2:] / 2), 1)


def map_metrics(dets, gts, iouv):
    from oracle import icaf_oracle as oracle
    tp, conf, pcls, tcls = [], [], [], []
    for d, gt in zip(dets, gts):
        tp.append(oracle.match_predictions(d, gt, iouv)); conf.append(d[:, 4]); pcls.append(d[:, 5]); tcls.append(gt[:, 0])
    ap, _ = oracle.ap_per_class(np.concatenate(tp), np.concatenate(conf), np.concatenate(pcls), np.concatenate(tcls))
    return 100.0 * float(ap[:, 0].mean()), 100.0 * float(ap.mean())


# ---- mAP@50 on a detection set with SEPARATED scores ---------------------------------------------------------------
# Random synthetic weights give thousands of near-equal scores per image (and, with i.i.d. noise images, features that barely
# depend on the image at all): mAP on such a set measures rank noise, not the implementation.  No trained checkpoint exists in
# the container, so the separation is PLANTED in the weights, the way a trained detector has it:
#   * "objects": a fixed +-A pattern added to the P4 DMFF position embedding (a learned (1, N, C) parameter of the reference,
#     models/common.py:773-774) at six token positions - a localised, image-independent cause that travels through the
#     cross-attention block, the bilinear merge and the whole PANet head;
#   * an objectness read-out FITTED to it: ridge regression of Detect's objectness row (one anchor per level; the other anchors
#     are switched off by their bias) on the fp32 oracle's head features, target +1 at the planted cells / -1 elsewhere, times a
#     gain - background cells end near conf 0.003, planted cells at 0.3-0.85, a gap of ~20x the bf16 noise of the logit;
#   * box regression rows damped (x 0.25) so that boxes stay near their anchors (the synthetic rows otherwise give 3 x 2 pixel
#     boxes, for which the 4-pixel grid of a bf16 coordinate near 640 already breaks IoU 0.5 in the REFERENCE's own bf16 mode).
# Ground truth = the fp32 oracle's detections above conf 0.25 plus 25 % labels that nothing detects (so that AP is not pinned
# at 100): every implementation that finds the same objects with boxes within IoU 0.5 and keeps them ranked above the
# background scores the same AP; a lost object costs 1 / (#objects) ~ 0.2 points.
PLANT_TOKENS = ((3, 4), (3, 11), (8, 8), (12, 3), (12, 12), (6, 14))     # of the 16 x 16 P4 token grid
PLANT_AMPLITUDE, PLANT_RIDGE, PLANT_GAIN, PLANT_ANCHOR, PLANT_REG_SCALE, PLANT_FIT_IMAGES = 2.0, 1e-3, 8.0, 2, 0.25, 4
# Round 4: the recipe above plants at P4 and reads out on the largest anchor of every level — 30-60 pixel boxes,
# for which no 16-bit box error can cross IoU 0.5.  "p3_small" plants at the P3 DMFF block (20 x 20 token grid), reads out ONLY on
# the P3 Detect level with its SMALLEST anchor (10 x 13 pixels) and damps the regression rows harder, so that every object is an
# 8-16 pixel box: two pixels of coordinate error then cost a third of the IoU, and the metric does depend on localisation (the
# reference's own bf16 mode, which decodes boxes INTO a bf16 tensor — a 2-4 pixel grid beyond x = 256 — shows it).
RECIPES = {
    "p4": dict(block=1, tokens=PLANT_TOKENS, anchor=PLANT_ANCHOR, reg_scale=PLANT_REG_SCALE, levels=(0, 1, 2)),
    "p3_small": dict(block=0, tokens=((2, 3), (3, 14), (6, 8), (9, 17), (10, 4), (13, 11), (16, 15), (17, 2), (5, 18), (14, 7)),
                     anchor=0, reg_scale=0.04, levels=(0,)),
}


def planted_detector(cfg, fsd, rgb, ir, seed, recipe="p4"):
    """Modify the fused state_dict `fsd` in place (see above); rgb / ir are the evaluation inputs, the first few of them
    are used for the fit.  Returns the fitted read-out's separation statistics."""
    from oracle import icaf_oracle as oracle
    rc = RECIPES[recipe]
    nc = cfg["nc"]
    no = nc + 5
    rows = cfg["backbone"] + cfg["head"]
    det_i = len(rows) - 1
    det_from = rows[-1][0]
    p4 = [i for i, r in enumerate(rows) if r[2] == "TransformerFusionBlock"][rc["block"]]
    va, ha = rows[p4][3][1], rows[p4][3][2]
    g = np.random.default_rng([seed, 0x91A47])
    C = fsd[f"model.{p4}.pos_emb_vis"].shape[2]
    u = torch.from_numpy(g.choice([-1.0, 1.0], C).astype(np.float32)) * PLANT_AMPLITUDE
    for k in (f"model.{p4}.pos_emb_vis", f"model.{p4}.pos_emb_ir"):
        for ty, tx in rc["tokens"]:
            fsd[k][0, ty * ha + tx] += u
    n = min(PLANT_FIT_IMAGES, rgb.shape[0])
    _, outs = oracle.OracleModel(cfg, fsd).forward(rgb[:n], ir[:n], keep_layers=True)
    stats = []
    for l, f in enumerate(det_from):
        wt, bt = fsd[f"model.{det_i}.m.{l}.weight"], fsd[f"model.{det_i}.m.{l}.bias"]
        if l not in rc["levels"]:                          # this level takes no part: every anchor's objectness switched off
            for an in range(3):
                wt[an * no + 4] = 0.0
                bt[an * no + 4] = -30.0
            continue
        feat = outs[f].numpy()
        bn, cc, ny, nx = feat.shape
        lab = -np.ones((ny, nx), np.int8)
        yy, xx = np.mgrid[0:ny, 0:nx]
        for ty, tx in rc["tokens"]:
            cy, cx = (ty + 0.5) * ny / va - 0.5, (tx + 0.5) * nx / ha - 0.5
            d = np.maximum(np.abs(yy - cy) * va / ny, np.abs(xx - cx) * ha / nx)       # distance in token units
            lab[d < 1.2] = 0                                                            # transition band: not fitted
            lab[d < 0.24 + 0.3 * va / ny] = 1
        y = np.tile(lab[None], (bn, 1, 1)).reshape(-1)
        a = np.concatenate((feat.transpose(0, 2, 3, 1).reshape(-1, cc), np.ones((y.size, 1), np.float32)), 1).astype(np.float64)
        sel = y != 0
        w = np.linalg.solve(a[sel].T @ a[sel] + PLANT_RIDGE * sel.sum() * np.eye(cc + 1), a[sel].T @ y[sel].astype(np.float64))
        sc = a @ w
        stats.append({"level": l, "planted_cells_min": float(sc[y == 1].min()), "background_max": float(sc[y == -1].max())})
        for an in range(3):
            c = an * no + 4
            if an == rc["anchor"]:
                wt[c, :, 0, 0] = torch.from_numpy((PLANT_GAIN * w[:-1]).astype(np.float32))
                bt[c] = float(PLANT_GAIN * w[-1])
            else:
                wt[c] = 0.0
                bt[c] = -30.0
            wt[an * no:an * no + 4] *= rc["reg_scale"]
            bt[an * no:an * no + 4] *= rc["reg_scale"]
            bt[an * no + 5:an * no + no] += 2.0
    return stats


def planted_case(yaml_name, B, H, W, seed, recipe="p4"):
    """(cfg, fused planted state_dict, rgb, ir, fit statistics) - CPU only."""
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_images, synth_state_dict
    cfg = load_cfg(yaml_name)
    m = Model(cfg).eval()
    m.load_state_dict(synth_state_dict(m, seed))
    m.fuse()
    fsd = {k: v.clone() for k, v in m.state_dict().items()}
    rgb, ir = synth_images(B, H, W, seed=seed)
    stats = planted_detector(cfg, fsd, rgb, ir, seed, recipe)
    return cfg, fsd, rgb, ir, stats


def planted_ground_truth(zr, nc):
    """Labels (cls, x1, y1, x2, y2) per image: the fp32 oracle's detections above conf 0.25 (NMS at the evaluation's IoU 0.5, so that
    the conf-0.001 detection list contains exactly these boxes) + one undetectable 2-pixel label per four objects."""
    from oracle import icaf_oracle as oracle
    strong = oracle.non_max_suppression(zr, 0.25, 0.5, multi_label=nc > 1)
    ghost = np.array([[0, 1, 1, 3, 3]], np.float32)
    objects = [np.concatenate((d[:, 5:6], d[:, :4]), 1).astype(np.float32) for d in strong]
    return objects, [np.concatenate((o, np.tile(ghost, (max(1, len(o) // 4), 1)))) for o in objects]


def measure_map(dtype, B=16, H=640, W=640, seed=6, yaml_name="yolov5s_Transfusion_kaist.yaml", hip=True, recipe="p4"):
    """mAP@50 / mAP@50:95 (percent; test.py's protocol: conf 0.001, IoU 0.5, multi-label when nc > 1, same ap_per_class) of the
    fp32 oracle, of the reference evaluated in `dtype` (the oracle by torch in that type) and - hip=True, GPU box - of the HIP
    path in `dtype`, on the planted detector above.  north_star: |delta mAP@50| <= 0.1."""
    from oracle import icaf_oracle as oracle
    _cpu_threads()
    cfg, fsd, rgb, ir, stats = planted_case(yaml_name, B, H, W, seed, recipe)
    nc = cfg["nc"]
    ml = nc > 1
    zr = oracle.OracleModel(cfg, fsd).forward(rgb, ir)[0].numpy()
    z16 = oracle.OracleModel(cfg, fsd, dtype=DT[dtype]).forward(rgb, ir)[0].float().numpy()      # the reference in the same 16-bit type
    objects, gts = planted_ground_truth(zr, nc)
    iouv = np.linspace(0.5, 0.95, 10)
    dets_r = oracle.non_max_suppression(zr, 0.001, 0.5, multi_label=ml)
    dets_16 = oracle.non_max_suppression(z16, 0.001, 0.5, multi_label=ml)
    conf = zr[..., 4] * zr[..., 5:].max(-1)
    b50, b = map_metrics(dets_r, gts, iouv)
    c50, c = map_metrics(dets_16, gts, iouv)
    wh = np.concatenate([o[:, 3:5] - o[:, 1:3] for o in objects]) if sum(len(o) for o in objects) else np.zeros((0, 2))
    out = {"dtype": dtype, "yaml": yaml_name, "images": B, "height": H, "width": W, "recipe": f"planted objects + fitted objectness read-out (tools/parity16.py, RECIPES[{recipe!r}])",
           "object_box_px": {"w_median": round(float(np.median(wh[:, 0])), 2), "h_median": round(float(np.median(wh[:, 1])), 2),
                             "min": round(float(wh.min()), 2), "max": round(float(wh.max()), 2)} if len(wh) else None,
           "fit": stats, "objects": int(sum(len(g) for g in objects)), "labels": int(sum(len(g) for g in gts)),
           "detections_oracle": int(sum(len(d) for d in dets_r)),
           "rows_conf_above_0.25": int((conf > 0.25).sum()), "rows_conf_0.05_to_0.25": int(((conf > 0.05) & (conf <= 0.25)).sum()),
           "rows_conf_0.001_to_0.05": int(((conf > 0.001) & (conf <= 0.05)).sum()),
           "map50_oracle_fp32": round(b50, 4), "map_oracle_fp32": round(b, 4),
           "map50_reference16": round(c50, 4), "map50_delta_reference16": round(c50 - b50, 4),
           "map_reference16": round(c, 4), "map_delta_reference16": round(c - b, 4)}
    if hip:
        from icafusion_amd.models.yolo import Model
        from icafusion_amd.utils.general import non_max_suppression
        m = Model(cfg).eval().fuse()
        m.load_state_dict(fsd)
        m = m.to("cuda:0")
        m.compute_dtype = DT[dtype]
        zg = m(rgb.cuda(), ir.cuda())[0].float()
        dets_g = [d.cpu().numpy() for d in non_max_suppression(zg, 0.001, 0.5, multi_label=ml)]
        a50, a = map_metrics(dets_g, gts, iouv)
        out.update({"detections_hip": int(sum(len(d) for d in dets_g)), "map50_hip16": round(a50, 4), "map50_delta": round(a50 - b50, 4),
                    "map_hip16": round(a, 4), "map_delta": round(a - b, 4)})
        del m
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_16bit.json"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    names = [n for n in CONFIGS if not args.only or any(n.startswith(p) for p in args.only.split(","))]
    res = {"method": __doc__.split("\n\n")[1], "device": torch.cuda.get_device_name(0), "host_threads": torch.get_num_threads(),
           "configs": [], "map50": []}
    for n in names:
        rec = measure(n)
        print(json.dumps(rec), flush=True)
        res["configs"].append(rec)
    if not args.only:
        for dt in ("bf16", "f16"):
            rec = measure_map(dt)
            print(json.dumps(rec), flush=True)
            res["map50"].append(rec)
        res["map50_small_objects"] = []
        for dt in ("bf16", "f16"):
            rec = measure_map(dt, recipe="p3_small")
            print(json.dumps(rec), flush=True)
            res["map50_small_objects"].append(rec)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
