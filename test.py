#!/usr/bin/env python3
"""Validation loop (mAP@0.5, mAP@0.5:0.95) of the two-stream detector on MI355X — the reference's test.py:23-330 reduced
to its metric path: paired RGB/IR loader -> forward -> NMS(multi_label) -> per-image TP matching at 10 IoU thresholds ->
ap_per_class.  `test(...)` keeps the reference's return value ((mp, mr, map50, map, 0, 0, 0), maps, times).

    python test.py --data data/multispectral/kaist.yaml --weights best.pt --batch-size 32 --img-size 640
    python test.py --data ... --cfg models/transformer/yolov5s_Transfusion_kaist.yaml      (synthetic weights: plumbing)

Protocol as the reference's: rectangular batches with pad 0.5 (test.py:100 — KAIST's 512x640 frames become 544x672
batches), conf 0.001 / IoU 0.5 multi-label NMS, boxes mapped back to native image space before matching.  Every distinct
batch shape compiles its own execution plan (hipGraph); Model keeps them in a byte-capped LRU (Model.plan_cache_bytes).
`--save-txt` / `--save-conf` / `--save-json` leave the reference's result files (per-image `frame,x,y,w,h[,conf]` lines + `result.txt`,
the input of the KAIST miss-rate evaluator; `<weights>_predictions.json`) under `--project/--name` (icafusion_amd/utils/results.py);
`--task speed` runs at conf 0.25 / IoU 0.45 as the reference does.  Not carried over (all outside the metric): plots / wandb /
`--save-hybrid` auto-labelling / `--task study` / the pycocotools call / the MR evaluator itself (the reference's MR call site is
disabled and returns zeros, test.py:260-285); `--augment` raises (test-time augmentation is not built)."""
import argparse
import os
import time

import numpy as np
import torch
import yaml

from icafusion_amd.models.experimental import attempt_load
from icafusion_amd.models.yolo import Model
from icafusion_amd.utils.datasets import create_dataloader_rgb_ir
from icafusion_amd import ops
from icafusion_amd.utils.general import check_img_size, increment_path, nms_device, scale_coords, xywh2xyxy
from icafusion_amd.utils.results import ResultWriter, label_listing
from icafusion_amd.utils.metrics import ap_per_class
from icafusion_amd.utils.torch_utils import select_device, time_synchronized


def summarize(stats, nc, names, seen, verbose=False):
    """Statistics -> (mp, mr, map50, map, maps) + the reference's result table (test.py:287-313): for one class the columns are
    TP / FP / FN / F1 / P / R / mAP@.5 / mAP@.5:.95, otherwise P / R / mAP@.5 / mAP@.75 / mAP@.5:.95 with one row per class when
    `verbose`.  stats: list of (correct (n, 10) bool, conf (n,), pcls (n,), tcls list) per image."""
    mp = mr = map50 = map75 = map_ = 0.0
    tp = fp = fn = f1 = np.zeros(1)
    p = r = np.zeros(0)
    ap, ap_class, nt = np.zeros((0, 10)), np.zeros(0, int), np.zeros(nc, int)
    if stats:
        cat = [np.concatenate([np.asarray(s[k]) for s in stats], 0) for k in range(4)]
        if len(cat[0]) and cat[0].any():
            tp, fp, fn, p, r, ap, f1, ap_class = ap_per_class(cat[0], cat[1], cat[2], cat[3])
            mp, mr, map50, map75, map_ = p.mean(), r.mean(), ap[:, 0].mean(), ap[:, 5].mean(), ap.mean()
        nt = np.bincount(cat[3].astype(np.int64), minlength=nc)
    lines = []
    if nc == 1:
        lines.append(("%20s" + "%12s" * 10) % ("Class", "Images", "Labels", "TP", "FP", "FN", "F1", "P", "R", "mAP@.5", "mAP@.5:.95"))
        lines.append(("%20s" + "%12i" * 2 + "%12.4g" * 8) % ("all", seen, nt.sum(), np.sum(tp), np.sum(fp), np.sum(fn), np.mean(f1), mp, mr,
                                                           map50, map_))
    else:
        pf = "%20s" + "%12i" * 2 + "%12.3g" * 5
        lines.append(("%20s" + "%12s" * 7) % ("Class", "Images", "Labels", "P", "R", "mAP@.5", "mAP@.75", "mAP@.5:.95"))
        lines.append(pf % ("all", seen, nt.sum(), mp, mr, map50, map75, map_))
        if verbose:
            for i, c in enumerate(ap_class):
                lines.append(pf % (names[c], seen, nt[c], p[i], r[i], ap[i, 0], ap[i, 5], ap[i].mean()))
    maps = np.zeros(nc) + map_
    for i, c in enumerate(ap_class):
        maps[c] = ap[i].mean()
    return (mp, mr, map50, map_), maps, lines


@torch.no_grad()
def test(data, weights=None, batch_size=32, imgsz=640, conf_thres=0.001, iou_thres=0.5, single_cls=False, model=None,
         dataloader=None, device="0", compute_dtype=None, cfg=None, verbose=False, save_json=False, save_txt=False, save_conf=True,
         save_dir=None, augment=False):
    if augment:
        raise NotImplementedError("test-time augmentation (models/yolo_test.py:116-132) is outside the inference hot path")
    if isinstance(data, str):
        with open(data) as f:
            data = yaml.safe_load(f)
    nc = 1 if single_cls else int(data["nc"])
    if model is None:
        dev = select_device(device)
        if weights:
            model = attempt_load(weights, map_location="cpu")
        else:
            from icafusion_amd.synth import synth_state_dict
            model = Model(cfg, nc=nc).eval()
            model.load_state_dict(synth_state_dict(model, seed=0))
            model = model.fuse().eval()
        model = model.to(dev)
        model.compute_dtype = compute_dtype
        model.use_graph = True
    dev = next(model.parameters()).device
    if dataloader is None:
        gs = int(max(float(model.stride.max()), 32))                      # grid size = max stride (test.py:68)
        # the reference passes `opt` (test.py:100), whose single_cls makes the dataset zero the label classes (utils/datasets.py:465-466)
        dataloader = create_dataloader_rgb_ir(data["val_rgb"], data["val_ir"], imgsz, batch_size, gs, argparse.Namespace(single_cls=single_cls),
                                              pad=0.5, rect=True)[0]
    writer = None
    if save_txt or save_json:                                           # result files of the reference (utils/results.py)
        listing = label_listing(os.path.dirname(dataloader.dataset.label_files[0])) if save_txt else None
        writer = ResultWriter(save_dir if save_dir is not None else increment_path("runs/test/exp"), save_txt=save_txt, save_conf=save_conf,
                              save_json=save_json, label_names=listing, weights=weights)
    iouv = np.linspace(0.5, 0.95, 10)
    names = data.get("names", [str(i) for i in range(nc)])
    stats, seen, t_inf, t_nms = [], 0, 0.0, 0.0
    for img, targets, paths, shapes in dataloader:
        img = img.to(dev, non_blocking=True)                     # uint8 (B, 6, H, W): cat(rgb, ir), test.py:116-123
        nb, _, height, width = img.shape
        t = time_synchronized()
        out = model.forward_u8(img)[0]
        t_inf += time_synchronized() - t
        targets = targets.clone()
        targets[:, 2:] *= torch.tensor([width, height, width, height])
        t = time_synchronized()
        det, count, _ = nms_device(out, conf_thres, iou_thres, multi_label=True, agnostic=single_cls)
        t_nms += time_synchronized() - t
        # statistics per image (reference test.py:144-230) on the device: labels go up in native image space, one kernel
        # maps the detections there (scale_coords + clip_coords) and matches them; only flags / conf / cls come back
        if single_cls:
            det[..., 5] = 0
        lab_rows, off, scale, tcls_all = [], [0], [], []
        for si in range(nb):
            labels = targets[targets[:, 0] == si, 1:]
            tcls_all.append(labels[:, 0].tolist())
            tbox = xywh2xyxy(labels[:, 1:5])
            scale_coords(img[si].shape[1:], tbox, shapes[si][0], shapes[si][1])               # native-space labels
            lab_rows.append(torch.cat((labels[:, :1], tbox), 1))
            off.append(off[-1] + len(labels))
            (h0, w0), ((gain, _), (padw, padh)) = shapes[si][0], shapes[si][1]
            scale.append([gain, padw, padh, w0, h0])
        predn = torch.zeros((nb, det.shape[1], 4), dtype=torch.float32, device=dev) if writer else None     # native-space boxes
        correct = ops.match_predictions(det, count, torch.cat(lab_rows).float().contiguous().to(dev),
                                        torch.tensor(off, dtype=torch.int32, device=dev), torch.from_numpy(iouv.astype(np.float32)).to(dev),
                                        scale=torch.tensor(scale, dtype=torch.float32, device=dev), predn=predn)
        correct, cc, count = correct.cpu().numpy().astype(bool), det[..., 4:6].cpu().numpy(), count.cpu().numpy()
        predn = predn.cpu().numpy() if writer else None
        for si in range(nb):
            n, tcls = int(count[si]), tcls_all[si]
            seen += 1
            if writer and n:
                writer.add(paths[si], predn[si, :n], cc[si, :n, 0], cc[si, :n, 1])
            if n == 0:
                if tcls:
                    stats.append((np.zeros((0, 10), bool), np.zeros(0), np.zeros(0), tcls))
                continue
            stats.append((correct[si, :n] if tcls else np.zeros((n, 10), bool), cc[si, :n, 0], cc[si, :n, 1], tcls))
    if writer:
        result_txt, pred_json = writer.close()
        for f in (result_txt, pred_json):
            if f is not None:
                print(f"saved {f}")
    (mp, mr, map50, map_), maps, lines = summarize(stats, nc, names, seen, verbose)
    print("\n".join(lines))
    tt = tuple(x / max(seen, 1) * 1e3 for x in (t_inf, t_nms, t_inf + t_nms)) + (imgsz, imgsz, batch_size)
    print("Speed: %.1f/%.1f/%.1f ms inference/NMS/total per %gx%g image at batch-size %g" % tt)
    return (mp, mr, map50, map_, 0.0, 0.0, 0.0), maps, tt


if __name__ == "__main__":
    ap_ = argparse.ArgumentParser(prog="test.py")
    ap_.add_argument("--weights", nargs="+", type=str, default=None)
    ap_.add_argument("--cfg", type=str, default="models/transformer/yolov5s_Transfusion_kaist.yaml")
    ap_.add_argument("--data", type=str, default="data/multispectral/kaist.yaml")
    ap_.add_argument("--batch-size", type=int, default=32)
    ap_.add_argument("--img-size", type=int, default=640)
    ap_.add_argument("--conf-thres", type=float, default=0.001)
    ap_.add_argument("--iou-thres", type=float, default=0.5)
    ap_.add_argument("--device", default="0")
    ap_.add_argument("--single-cls", action="store_true")
    ap_.add_argument("--half", action="store_true")
    ap_.add_argument("--bf16", action="store_true")
    ap_.add_argument("--verbose", action="store_true")
    ap_.add_argument("--task", default="val", help="val / test / train: one validation pass (the reference reads val_rgb / val_ir for all three); "
                                                   "speed: conf 0.25 / IoU 0.45, nothing saved")
    ap_.add_argument("--augment", action="store_true", help="test-time augmentation: not built, raises")
    ap_.add_argument("--save-txt", action="store_true", help="per-image result lines + result.txt under <project>/<name>/labels")
    ap_.add_argument("--save-conf", action="store_true", help="append the confidence to every --save-txt line")
    ap_.add_argument("--save-json", action="store_true", help="<project>/<name>/<weights>_predictions.json")
    ap_.add_argument("--save-hybrid", action="store_true", help="auto-labelling: not built, raises")
    ap_.add_argument("--project", default="runs/test")
    ap_.add_argument("--name", default="exp")
    ap_.add_argument("--exist-ok", action="store_true")
    o = ap_.parse_args()
    print(o)
    if o.save_hybrid:
        raise NotImplementedError("--save-hybrid (label + prediction auto-labelling, test.py:134-135) is outside the inference hot path")
    if o.task not in ("val", "test", "train", "speed"):
        raise NotImplementedError(f"--task {o.task}: only val / test / train / speed are built (study = image-size sweep + plots)")
    dtype = torch.float16 if o.half else torch.bfloat16 if o.bf16 else None
    imgsz = check_img_size(o.img_size, 32)
    if o.task == "speed":                                                   # test.py:420-422
        for w in (o.weights or [None]):
            test(o.data, [w] if w else None, o.batch_size, imgsz, 0.25, 0.45, o.single_cls, device=o.device, compute_dtype=dtype, cfg=o.cfg)
    else:
        save_dir = increment_path(os.path.join(o.project, o.name), exist_ok=o.exist_ok) if (o.save_txt or o.save_json) else None
        test(o.data, o.weights, o.batch_size, imgsz, o.conf_thres, o.iou_thres, o.single_cls, device=o.device, compute_dtype=dtype, cfg=o.cfg,
             verbose=o.verbose, save_json=o.save_json, save_txt=o.save_txt, save_conf=o.save_conf, save_dir=save_dir, augment=o.augment)
