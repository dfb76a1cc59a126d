#!/usr/bin/env python3
"""Paired-folder RGB / IR inference on MI355X — the front end of the reference's detect_twostream.py (same flags for the
image-folder path: `--source1` visible images, `--source2` infrared images, zipped in sorted order, :56-66).

    python detect_twostream.py --weights best.pt --source1 visible/ --source2 infrared/ [--save-txt] [--half | --bf16]
    python detect_twostream.py --cfg models/transformer/yolov5s_Transfusion_kaist.yaml --source1 ... --source2 ...
                               (no trained ICAFusion weights ship with the reference: --cfg builds the network with
                                deterministic synthetic weights so that the whole path can be exercised)

Per pair: letterbox to --img-size (utils/datasets.py), uint8 -> device, one forward (hipGraph replay) + device NMS,
boxes scaled back to the original image, optional YOLO-format txt / annotated images.  The per-frame
`Done. (…s, …Hz)` and final `Average Speed` lines are the reference's (:160,198).  Webcam / video sources, the
second-stage classifier and --view-img need OpenCV and are out of scope."""
import argparse
import time
from pathlib import Path

import numpy as np
import torch

from icafusion_amd.models.experimental import attempt_load
from icafusion_amd.models.yolo import Model
from icafusion_amd.utils.datasets import LoadImages, imwrite_bgr
from icafusion_amd.utils.general import increment_path, non_max_suppression, scale_coords, xyxy2xywh
from icafusion_amd.utils.torch_utils import select_device, time_synchronized


def draw_boxes(img_bgr, det, names, thickness=2, hide_labels=False):
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.ascontiguousarray(img_bgr[:, :, ::-1]))
    d = ImageDraw.Draw(im)
    for *xyxy, conf, cls in det.tolist():
        c = int(cls)
        color = ((37 * c + 60) % 256, (91 * c + 160) % 256, (151 * c + 30) % 256)
        d.rectangle(xyxy, outline=color, width=thickness)
        if not hide_labels:
            d.text((xyxy[0] + 2, max(xyxy[1] - 11, 0)), names[c], fill=color)
    return np.asarray(im)[:, :, ::-1]


def load_model(opt, device):
    if opt.weights:
        model = attempt_load(opt.weights, map_location="cpu")
    else:
        from icafusion_amd.synth import synth_state_dict
        model = Model(opt.cfg).eval()
        model.load_state_dict(synth_state_dict(model, seed=0))
        model = model.fuse().eval()
    model = model.to(device)
    model.compute_dtype = torch.float16 if opt.half else torch.bfloat16 if opt.bf16 else None
    model.use_graph = True
    return model


@torch.no_grad()
def detect(opt):
    if getattr(opt, "augment", False):
        raise NotImplementedError("test-time augmentation (models/yolo_test.py:116-132) is outside the inference hot path")
    device = select_device(opt.device)
    save_img = not opt.nosave
    save_dir = increment_path(Path(opt.project) / opt.name, exist_ok=opt.exist_ok)
    (save_dir / "labels" if opt.save_txt else save_dir).mkdir(parents=True, exist_ok=True)
    model = load_model(opt, device)
    stride = int(model.stride.max())
    names = model.names
    dataset, dataset2 = LoadImages(opt.source1, opt.img_size, stride), LoadImages(opt.source2, opt.img_size, stride)
    if len(dataset) != len(dataset2):
        raise ValueError(f"{len(dataset)} visible images vs {len(dataset2)} infrared images")
    t0, img_num, fps_sum = time.time(), 0, 0.0
    for (path, img, im0, _), (path2, img2, im0_, _) in zip(dataset, dataset2):
        img6 = torch.from_numpy(np.concatenate((img, img2), 0)).unsqueeze(0).to(device)     # uint8 (1, 6, H, W)
        t1 = time_synchronized()
        pred = model.forward_u8(img6)[0]           # /255, RGB/IR split and the cast happen in the staging kernel
        pred = non_max_suppression(pred, opt.conf_thres, opt.iou_thres, classes=opt.classes, agnostic=opt.agnostic_nms)
        t2 = time_synchronized()
        det = pred[0]
        p = Path(path)
        s = "%gx%g " % tuple(img6.shape[2:])
        if len(det):
            det[:, :4] = scale_coords(img6.shape[2:], det[:, :4], im0.shape).round()
            for c in det[:, -1].unique():
                n = int((det[:, -1] == c).sum())
                s += f"{n} {names[int(c)]}{'s' * (n > 1)}, "
            if opt.save_txt:
                gn = torch.tensor(im0.shape)[[1, 0, 1, 0]].to(det.device)
                with open(save_dir / "labels" / (p.stem + ".txt"), "a") as f:
                    for *xyxy, conf, cls in reversed(det.tolist()):
                        xywh = (xyxy2xywh(torch.tensor(xyxy).view(1, 4)) / gn.cpu()).view(-1).tolist()
                        line = (cls, *xywh, conf) if opt.save_conf else (cls, *xywh)
                        f.write(("%g " * len(line)).rstrip() % line + "\n")
        if save_img:
            d = det.cpu().numpy() if len(det) else np.zeros((0, 6), np.float32)
            imwrite_bgr(str(save_dir / (p.stem + "_rgb" + p.suffix)), draw_boxes(im0, d, names, opt.line_thickness, opt.hide_labels))
            imwrite_bgr(str(save_dir / (p.stem + "_ir" + p.suffix)), draw_boxes(im0_, d, names, opt.line_thickness, opt.hide_labels))
        print(f"image {img_num + 1}/{len(dataset)} {path}: {s}Done. ({t2 - t1:.6f}s, {1 / (t2 - t1):.6f}Hz)")
        img_num += 1
        fps_sum += 1 / (t2 - t1)
    if opt.save_txt or save_img:
        print(f"Results saved to {save_dir}")
    print(f"Done. ({time.time() - t0:.3f}s)")
    print(f"Average Speed: {fps_sum / max(img_num, 1):.6f}Hz")
    return save_dir


def parse_opt(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", nargs="+", type=str, default=None, help="model.pt path(s) (pickled reference checkpoint)")
    ap.add_argument("--cfg", type=str, default="models/transformer/yolov5s_Transfusion_kaist.yaml",
                    help="model yaml used with synthetic weights when --weights is not given")
    ap.add_argument("--source1", type=str, required=True, help="visible images: file / folder / glob")
    ap.add_argument("--source2", type=str, required=True, help="infrared images: file / folder / glob")
    ap.add_argument("--img-size", type=int, default=640)
    ap.add_argument("--conf-thres", type=float, default=0.1)
    ap.add_argument("--iou-thres", type=float, default=0.5)
    ap.add_argument("--device", default="0")
    ap.add_argument("--half", action="store_true", help="fp16 kernels (the reference's GPU default, :33,40)")
    ap.add_argument("--bf16", action="store_true", help="bf16 kernels")
    ap.add_argument("--save-txt", action="store_true")
    ap.add_argument("--save-conf", action="store_true")
    ap.add_argument("--nosave", action="store_true")
    ap.add_argument("--classes", nargs="+", type=int)
    ap.add_argument("--agnostic-nms", action="store_true")
    ap.add_argument("--project", default="runs/detect")
    ap.add_argument("--name", default="exp")
    ap.add_argument("--exist-ok", action="store_true")
    ap.add_argument("--line-thickness", default=2, type=int)
    ap.add_argument("--hide-labels", default=False, action="store_true")
    ap.add_argument("--hide-conf", default=True, action="store_true", help="accepted for the reference's command lines: its default is True and "
                    "cannot be switched off (:224), i.e. boxes carry the class name only — which is what is drawn here")
    ap.add_argument("--augment", action="store_true", help="test-time augmentation: not built, raises")
    return ap.parse_args(argv)


if __name__ == "__main__":
    opt = parse_opt()
    print(opt)
    detect(opt)
