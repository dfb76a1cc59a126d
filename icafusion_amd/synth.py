"""Deterministic synthetic weights and inputs.

There are no trained ICAFusion checkpoints or datasets in the container (they sit behind external links in the
reference README), so parity, smoke and bench runs all use a reproducible pseudo-random fill of the model's
``state_dict``.  The fill depends only on (seed, key name, shape) and uses numpy's PCG64 stream, which is stable
across platforms and library versions, so the reference model in this container, the CPU oracle and the HIP model
on the GPU box all see bit-identical fp32 weights without shipping 100 MB fixtures.
"""
import re
import zlib
import numpy as np
import torch

_KEEP = ("anchors", "anchor_grid")
_DETECT = re.compile(r"\.m\.\d+\.(weight|bias)$")
_RESBN = re.compile(r"\.m\.\d+\.cv2\.bn\.weight$")   # last BN of a residual Bottleneck: small gain keeps deep stacks O(1)


def _rng(seed, key):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def synth_tensor(key, shape, dtype=torch.float32, seed=0):
    """Value for one state_dict entry; scale rules keep activations O(1) through ~40 layers."""
    g = _rng(seed, key)
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        a = g.uniform(0.6, 1.4, n)
    elif leaf == "running_mean":
        a = g.normal(0.0, 0.2, n)
    elif leaf in ("w1", "w2"):
        a = g.uniform(0.3, 0.7, n)
    elif "coefficient" in key and leaf == "bias":
        a = g.uniform(0.7, 1.3, n)
    elif leaf.startswith("pos_emb"):
        a = g.normal(0.0, 0.3, n)
    elif _DETECT.search(key):                   # Detect head 1x1 convs: wide logits so sigmoid outputs spread out
        a = g.normal(0.0, 3.0 / np.sqrt(shape[1]), n) if leaf == "weight" else g.normal(-1.0, 1.0, n)
    elif len(shape) == 4:                       # conv weight (Cout, Cin, kh, kw)
        fan_in = shape[1] * shape[2] * shape[3]
        a = g.normal(0.0, 1.35 / np.sqrt(fan_in), n)
    elif len(shape) == 2:                       # linear weight (out, in)
        a = g.normal(0.0, 1.0 / np.sqrt(shape[1]), n)
    elif _RESBN.search(key):
        a = g.uniform(0.15, 0.45, n)
    elif leaf == "weight":                      # BN / LN gain
        a = g.uniform(0.7, 1.3, n)
    elif leaf == "bias":
        a = g.normal(0.0, 0.1, n)
    else:
        a = g.normal(0.0, 0.1, n)
    return torch.from_numpy(a.astype(np.float32).reshape(shape)).to(dtype)


def synth_state_dict(module, seed=0):
    """Return a full replacement state_dict for ``module`` (buffers named in _KEEP are left untouched)."""
    out = {}
    for k, v in module.state_dict().items():
        if k.rsplit(".", 1)[-1] in _KEEP:
            out[k] = v.clone()
        else:
            out[k] = synth_tensor(k, v.shape, v.dtype if v.dtype.is_floating_point else torch.int64, seed)
    return out


def synth_images(batch, height, width, seed=0, quantize=True):
    """Paired RGB / IR inputs in the post-``/255`` domain (detect_twostream.py:74 in the reference).

    quantize=True draws uint8 pixels and divides by 255 so values match what the real pipeline feeds."""
    g = np.random.default_rng([seed, 0xC0FFEE, batch, height, width])
    if quantize:
        a = g.integers(0, 256, size=(2, batch, 3, height, width), dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    else:
        a = g.random(size=(2, batch, 3, height, width), dtype=np.float32)
    t = torch.from_numpy(a)
    return t[0].contiguous(), t[1].contiguous()


def synth_labels(batch, nc, seed=0, max_boxes=8):
    """Per-image ground-truth boxes (cls, x, y, w, h) normalised, for mAP plumbing checks."""
    g = np.random.default_rng([seed, 0x1ABE1, batch, nc])
    rows = []
    for b in range(batch):
        n = int(g.integers(1, max_boxes + 1))
        cls = g.integers(0, nc, n)
        wh = g.uniform(0.05, 0.4, (n, 2))
        xy = g.uniform(0.2, 0.8, (n, 2))
        for i in range(n):
            rows.append([b, cls[i], xy[i, 0], xy[i, 1], wh[i, 0], wh[i, 1]])
    return torch.tensor(rows, dtype=torch.float32)
