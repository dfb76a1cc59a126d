"""Detection metrics on the host (numpy), same entry points as the reference's utils/metrics.py."""
import numpy as np


def fitness(x):
    """Weighted combination used for model selection: only mAP@0.5 counts (reference utils/metrics.py:12-15)."""
    w = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    return (x[:, :8] * w).sum(1)


def compute_ap(recall, precision):
    """101-point interpolated AP of one precision/recall curve (reference utils/metrics.py:85-110)."""
    mrec = np.concatenate(([0.0], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    ap = np.trapezoid(np.interp(x, mrec, mpre), x) if hasattr(np, "trapezoid") else np.trapz(np.interp(x, mrec, mpre), x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir=".", names=()):
    """Per-class AP at every IoU threshold plus P/R/F1 at the best-F1 confidence (reference utils/metrics.py:18-82).
    Returns (tp, fp, fn, p, r, ap, f1, classes) like the reference; plotting is not provided."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes = np.unique(target_cls)
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    n_l = 0
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_l = (target_cls == c).sum()
        n_p = sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + 1e-16)
        r[ci] = np.interp(-px, -conf[sel], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p[ci] = np.interp(-px, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + 1e-16)
    i = f1.mean(0).argmax()
    tpn = (r * n_l).round()
    fn = n_l - tpn
    fp = (tpn / (p + 1e-16) - tpn).round()
    return tpn[:, i], fp[:, i], fn[:, i], p[:, i], r[:, i], ap, f1[:, i], unique_classes.astype("int32")


def match_predictions(det, labels, iouv):
    """TP flags of one image's detections at every IoU threshold (reference test.py:196-230).

    det: (n, 6) [x1, y1, x2, y2, conf, cls]; labels: (m, 5) [cls, x1, y1, x2, y2] in the same pixel space; iouv: (T,).
    Per class each prediction is paired with its best-IoU target; predictions are visited in index order and a target
    is claimed by the first prediction above iouv[0] that reaches it.  Returns bool (n, T)."""
    det, labels, iouv = np.asarray(det, np.float32), np.asarray(labels, np.float32), np.asarray(iouv, np.float32)
    correct = np.zeros((det.shape[0], iouv.shape[0]), bool)
    if not len(det) or not len(labels):
        return correct
    claimed_total = 0
    for cls in np.unique(labels[:, 0]):
        ti = np.flatnonzero(labels[:, 0] == cls)
        pi = np.flatnonzero(det[:, 5] == cls)
        if not len(pi):
            continue
        a, b = det[pi, :4], labels[ti, 1:5]
        inter = (np.minimum(a[:, None, 2:], b[None, :, 2:]) - np.maximum(a[:, None, :2], b[None, :, :2])).clip(0).prod(2)
        area_a, area_b = (a[:, 2:] - a[:, :2]).prod(1), (b[:, 2:] - b[:, :2]).prod(1)
        iou = inter / (area_a[:, None] + area_b[None, :] - inter)
        best, arg = iou.max(1), iou.argmax(1)
        claimed = set()
        for j in np.flatnonzero(best > iouv[0]):
            t = int(ti[arg[j]])
            if t in claimed:
                continue
            claimed.add(t)
            claimed_total += 1
            correct[pi[j]] = best[j] > iouv
            if claimed_total == len(labels):
                break
    return correct
