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
