"""Post-processing with the reference's function names (utils/general.py): NMS runs on the device in HIP kernels,
the small coordinate helpers are plain tensor code on whatever device their inputs live on."""
import math
import time

import numpy as np
import torch

from .. import ops

_RUNNERS = {}          # small LRU of NmsRunners for the stand-alone entry points (pipelines own theirs)
_MAX_RUNNERS = 4


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def check_img_size(img_size, s=32):
    """Round --img-size up to a multiple of the model's largest stride (reference utils/general.py:142-147)."""
    new_size = make_divisible(img_size, int(s))
    if new_size != img_size:
        print("WARNING: --img-size %g must be multiple of max stride %g, updating to %g" % (img_size, s, new_size))
    return new_size


def increment_path(path, exist_ok=False, sep="", mkdir=False):
    """runs/exp -> runs/exp2, runs/exp3, ... when the path exists (reference utils/general.py:705-719): the next free
    number after the highest `<name><sep><n>` sibling, 2 when there is none; a file keeps its suffix."""
    import glob
    import re
    from pathlib import Path
    path = Path(path)
    if path.exists() and not exist_ok:
        suffix = path.suffix
        path = path.with_suffix("")
        taken = [re.search(rf"%s{sep}(\d+)" % re.escape(path.stem), d) for d in glob.glob(f"{path}{sep}*")]
        nums = [int(m.group(1)) for m in taken if m]
        path = Path(f"{path}{sep}{max(nums) + 1 if nums else 2}{suffix}")
    folder = path if path.suffix == "" else path.parent
    if mkdir and not folder.exists():
        folder.mkdir(parents=True, exist_ok=True)
    return path


def xyxy2xywh2(x):
    """[x1, y1, x2, y2] -> [x1, y1, w, h] (top-left corner + size: the KAIST result-file layout, reference :312-319)."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xyxy2xywh(x):
    """[x1, y1, x2, y2] -> [cx, cy, w, h] (reference utils/general.py:322-329)."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    """[cx, cy, w, h] -> [x1, y1, x2, y2] (reference utils/general.py:332-339)."""
    y = x.clone() if isinstance(x, torch.Tensor) else np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, img_shape):
    """Clip xyxy boxes to (height, width) in place (reference utils/general.py:402-407)."""
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy boxes from the letterboxed network input back to the native image (reference :386-399)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


def box_iou(box1, box2):
    """Pairwise IoU of xyxy boxes, (N,4) x (M,4) -> (N,M) (reference utils/general.py:455-477)."""
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def nms_device(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
               max_det=300, max_nms=30000, max_wh=4096.0, stream_ptr=None, runner=None):
    """Device-resident NMS: returns (det (B,max_det,6), count (B,), keep_idx (B,max_det)) without any host sync.
    The three tensors are the runner's STATIC output buffers: with the default (shared, per-shape) runner they are
    overwritten by the next call of the same shape — consume or clone them first, or pass an own `runner`
    (ops.NmsRunner) as DetectionPipeline does for each of its two in-flight slots."""
    if not prediction.is_cuda:
        raise RuntimeError("non_max_suppression runs on the MI355X only (no CPU fallback; see oracle/ for the "
                           "CPU reference used by the tests)")
    pred = prediction.float().contiguous()
    B, rows, no = pred.shape
    nc = no - 5
    ml = bool(multi_label) and nc > 1
    if runner is None:
        key = (B, rows, nc, ml, max_det, pred.device)
        runner = _RUNNERS.pop(key, None)
        if runner is None:
            while len(_RUNNERS) >= _MAX_RUNNERS:          # rectangular validation batches meet many (B, rows): bound the workspaces
                _RUNNERS.pop(next(iter(_RUNNERS)))
            runner = ops.NmsRunner(B, rows, nc, pred.device, ml, max_det)
        _RUNNERS[key] = runner
    elif (runner.B, runner.rows, runner.nc, runner.multi_label, runner.max_det) != (B, rows, nc, ml, max_det):
        raise ValueError("nms_device: the runner was built for another (B, rows, nc, multi_label, max_det)")
    return runner.launch(pred, conf_thres, iou_thres, agnostic, classes, max_nms, max_wh, stream_ptr)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=()):
    """Drop-in for the reference's non_max_suppression (utils/general.py:518-607): list of (n,6) tensors
    [x1, y1, x2, y2, conf, cls] per image.  The 10 s wall-clock bail-out of the reference (:603-605) does not exist
    here; apriori `labels` (autolabelling) are outside the inference hot path."""
    if labels:
        raise NotImplementedError("apriori labels (autolabelling) are outside the inference hot path")
    det, count, _ = nms_device(prediction, conf_thres, iou_thres, classes, agnostic, multi_label)
    counts = count.tolist()                           # the one device->host sync of post-processing
    return [det[i, :n].clone() for i, n in enumerate(counts)]
