#!/usr/bin/env python3
"""Validation loop (mAP@0.5, mAP@0.5:0.95) of the two-stream detector on MI355X — the reference's test.py:23-330 reduced
to its metric path: paired RGB/IR loader -> forward -> NMS(multi_label) -> per-image TP matching at 10 IoU thresholds ->
ap_per_class.  `test(...)` keeps the reference's return value ((mp, mr, map50, map, 0, 0, 0), maps, times).

    python test.py --data data/multispectral/kaist.yaml --weights best.pt --batch-size 32 --img-size 640
    python test.py --data ... --cfg models/transformer/yolov5s_Transfusion_kaist.yaml      (synthetic weights: plumbing)

Protocol as the reference's: rectangular batches with pad 0.5 (test.py:100 — KAIST's 512x640 frames become 544x672
batches), conf 0.001 / IoU 0.5 multi-label NMS, boxes mapped back to native image space before matching.  Every distinct
batch shape compiles its own execution plan (hipGraph); Model keeps them in a byte-capped LRU (Model.plan_cache_bytes).
Not carried over (all outside the metric): plots / wandb / json / the MR evaluator (the reference's MR call site is disabled
and returns zeros, test.py:260-285)."""
import argparse
import time

import numpy as np
import torch
import yaml

from icafusion_amd.models.experimental import attempt_load
from icafusion_amd.models.yolo import Model
from icafusion_amd.utils.datasets import create_dataloader_rgb_ir
from icafusion_amd import ops
from icafusion_amd.utils.general import nms_device, scale_coords, xywh2xyxy
from icafusion_amd.utils.metrics import ap_per_class
from icafusion_amd.utils.torch_utils import select_device, time_synchronized


@torch.no_grad()
def test(data, weights=None, batch_size=32, imgsz=640, conf_thres=0.001, iou_thres=0.5, single_cls=False, model=None,
         dataloader=None, device="0", compute_dtype=None, cfg=None, verbose=False):
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
        dataloader = create_dataloader_rgb_ir(data["val_rgb"], data["val_ir"], imgsz, batch_size, gs, None, pad=0.5, rect=True)[0]
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
        correct = ops.match_predictions(det, count, torch.cat(lab_rows).float().contiguous().to(dev),
                                        torch.tensor(off, dtype=torch.int32, device=dev), torch.from_numpy(iouv.astype(np.float32)).to(dev),
                                        scale=torch.tensor(scale, dtype=torch.float32, device=dev))
        correct, cc, count = correct.cpu().numpy().astype(bool), det[..., 4:6].cpu().numpy(), count.cpu().numpy()
        for si in range(nb):
            n, tcls = int(count[si]), tcls_all[si]
            seen += 1
            if n == 0:
                if tcls:
                    stats.append((np.zeros((0, 10), bool), np.zeros(0), np.zeros(0), tcls))
                continue
            stats.append((correct[si, :n] if tcls else np.zeros((n, 10), bool), cc[si, :n, 0], cc[si, :n, 1], tcls))
    mp = mr = map50 = map_ = 0.0
    ap, ap_class, nt = np.zeros((0, 10)), np.zeros(0, int), np.zeros(nc, int)
    if stats:
        cat = [np.concatenate([np.asarray(s[k]) for s in stats], 0) for k in range(4)]
        if len(cat[0]) and cat[0].any():
            _, _, _, p, r, ap, _, ap_class = ap_per_class(cat[0], cat[1], cat[2], cat[3])
            mp, mr, map50, map_ = p.mean(), r.mean(), ap[:, 0].mean(), ap.mean()
        nt = np.bincount(cat[3].astype(np.int64), minlength=nc)
    print(("%20s" + "%12s" * 6) % ("Class", "Images", "Labels", "P", "R", "mAP@.5", "mAP@.5:.95"))
    print(("%20s" + "%12i" * 2 + "%12.3g" * 4) % ("all", seen, nt.sum(), mp, mr, map50, map_))
    if verbose and nc > 1:
        for i, c in enumerate(ap_class):
            print(("%20s" + "%12i" * 2 + "%12.3g" * 2) % (names[c], seen, nt[c], ap[i, 0], ap[i].mean()))
    tt = tuple(x / max(seen, 1) * 1e3 for x in (t_inf, t_nms, t_inf + t_nms)) + (imgsz, imgsz, batch_size)
    print("Speed: %.1f/%.1f/%.1f ms inference/NMS/total per %gx%g image at batch-size %g" % tt)
    maps = np.zeros(nc) + map_
    for i, c in enumerate(ap_class):
        maps[c] = ap[i].mean()
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
    o = ap_.parse_args()
    print(o)
    test(o.data, o.weights, o.batch_size, o.img_size, o.conf_thres, o.iou_thres, o.single_cls, device=o.device,
         compute_dtype=torch.float16 if o.half else torch.bfloat16 if o.bf16 else None, cfg=o.cfg, verbose=o.verbose)
