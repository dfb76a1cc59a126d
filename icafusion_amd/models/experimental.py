"""Checkpoint loading with the reference's entry point name (models/experimental.py:98-134)."""
import torch
import torch.nn as nn

from .common import Conv
from .yolo import Detect, Model


class Ensemble(nn.ModuleList):
    """Ensemble of models; two-stream forward, concatenated predictions (reference models/experimental.py:98-110)."""

    def forward(self, x, x2, augment=False):
        ys = [m(x, x2, augment)[0] for m in self]
        return torch.cat(ys, 1), None


def attempt_load(weights, map_location=None):
    """Load one or more pickled checkpoints ({'ema'|'model': Model}) -> fused eval model(s).

    Reference checkpoints are whole pickled Model objects (train.py:424-435); they unpickle against this repo's
    `models.yolo_test.Model` / `models.common.*` aliases, which resolve to the HIP-backed classes."""
    model = Ensemble()
    for w in weights if isinstance(weights, list) else [weights]:
        ckpt = torch.load(w, map_location=map_location, weights_only=False)
        m = ckpt["ema" if ckpt.get("ema") else "model"] if isinstance(ckpt, dict) else ckpt
        model.append(m.float().fuse().eval())
    for m in model.modules():
        if type(m) in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU, Detect, Model):
            m.inplace = True
        elif type(m) is Conv:
            m._non_persistent_buffers_set = set()
    if len(model) == 1:
        return model[-1]
    for k in ("names", "stride"):
        setattr(model, k, getattr(model[-1], k))
    return model
