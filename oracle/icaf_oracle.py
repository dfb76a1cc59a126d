"""CPU oracle for the ICAFusion hot path — TEST INFRASTRUCTURE ONLY.

This file restates, as plain fp32 tensor arithmetic on the CPU, the algorithm of the reference's two-stream
forward path (backbone -> DMFF -> PANet head -> Detect) and of its NMS post-processing.  It exists so that the
HIP kernels can be checked on a GPU box where /root/reference is not available.  It is NOT part of the product:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinning: every function here is checked against outputs of the real reference (imported read-only in the build
container by tests/golden/make_golden.py) through the fixtures in tests/golden/*.npz — see
tests/test_oracle_vs_golden.py.  The one exception is the greedy-NMS core: the reference delegates it to
torchvision.ops.nms (utils/general.py:591), torchvision is neither vendored nor installed, so the core follows
torchvision's documented semantics and is "parity unpinned"; the wrapper logic around it IS pinned (the
reference's non_max_suppression was run with this core injected).

Every function cites the reference file:line whose behaviour it restates (paths relative to the reference root).
Nothing is copied: the reference is an nn.Module tree, this is a functional interpreter over (cfg dict,
state_dict).
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3     # utils/torch_utils.py:150-152 (initialize_weights sets eps on every BatchNorm2d)
LN_EPS = 1e-5     # nn.LayerNorm default, models/common.py:625-626,701-702
STRIDES = (8.0, 16.0, 32.0)  # models/yolo_test.py:104


# --------------------------------------------------------------------------------------------------------------
# cfg -> layer list                                                   (restates models/yolo_test.py:216-302)
# --------------------------------------------------------------------------------------------------------------
def _round_up(x, d):
    return int(math.ceil(x / d) * d)      # utils/general.py make_divisible


def build_layers(cfg):
    """Return [(index, from, kind, params)] and per-layer output channels."""
    anchors, nc = cfg["anchors"], cfg["nc"]
    gd, gw = cfg["depth_multiple"], cfg["width_multiple"]
    na = len(anchors[0]) // 2
    no = na * (nc + 5)
    ch, layers = [], []
    for i, (f, n, kind, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = [nc if a == "nc" else anchors if a == "anchors" else (None if a == "None" else a) for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if kind in ("Conv", "C3", "SPPF"):
            c2 = args[0]
            c1 = 3 if (kind == "Conv" and args[0] == 64) else ch[f]     # models/yolo_test.py:240-244
            if c2 != no:
                c2 = _round_up(c2 * gw, 8)
            p = dict(c1=c1, c2=c2)
            if kind == "Conv":
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                pad = args[3] if len(args) > 3 and args[3] is not None else k // 2
                p.update(k=k, s=s, p=pad)
            elif kind == "C3":
                p.update(n=n, shortcut=args[1] if len(args) > 1 else True)
                n = 1
            else:
                p.update(k=args[1] if len(args) > 1 else 5)
        elif kind == "nn.Upsample":
            c2, p = ch[f], dict(scale=args[1], mode=args[2])
        elif kind == "Concat":
            c2, p = sum(ch[x] for x in f), dict(dim=args[0])
        elif kind == "TransformerFusionBlock":
            c2 = ch[f[0]]                                                # yaml's first arg is ignored (:284-286)
            kw = args[-1] if isinstance(args[-1], dict) else {}         # this repo's optional {loops_num: n} (not in the reference)
            pos = [a for a in args if not isinstance(a, dict)]
            p = dict(c=c2, va=pos[1], ha=pos[2], heads=pos[3] if len(pos) > 3 else 8, loops=kw.get("loops_num", 1))
        elif kind == "NiNfusion":                                       # models/yolo_test.py:280-283
            c1 = sum(ch[x] for x in f)
            c2, p = c1 // 2, dict(k=args[0], s=args[1])
        elif kind == "Add":                                             # models/yolo_test.py:266-268: args = [c2], so the
            c2 = ch[f[0]]                                               # "weight" of Add.__init__ is the CHANNEL COUNT
            p = dict(w=float(c2))
        elif kind == "Detect":
            c2, p = None, dict(nc=args[0], anchors=args[1], ch=[ch[x] for x in f])
        else:
            raise NotImplementedError(f"oracle does not cover module {kind} (out of §8 scope)")
        assert n == 1, "only C3 uses depth repeats on the hot path"
        layers.append((i, f, kind, p))
        ch.append(c2)
    return layers, ch


# --------------------------------------------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------------------------------------------
def conv_bn_silu(x, sd, pre, k, s, p, act=True):
    """models/common.py:48-60 (Conv = SiLU(BN(Conv2d no-bias))), BN in eval mode."""
    if pre + ".bn.weight" not in sd:          # fused checkpoint (utils/torch_utils.py:182-202): Conv.fuseforward = act(conv(x)), one op
        y = F.conv2d(x, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"], s, p)
        return F.silu(y) if act else y
    y = F.conv2d(x, sd[pre + ".conv.weight"], None, s, p)
    g, b = sd[pre + ".bn.weight"], sd[pre + ".bn.bias"]
    mu, var = sd[pre + ".bn.running_mean"], sd[pre + ".bn.running_var"]
    scale = g / torch.sqrt(var + BN_EPS)
    y = (y - mu[None, :, None, None]) * scale[None, :, None, None] + b[None, :, None, None]
    return F.silu(y) if act else y


def c3(x, sd, pre, n, shortcut):
    """models/common.py:216-227 with Bottleneck :184-194 (1x1 then 3x3, residual only when shortcut)."""
    a = conv_bn_silu(x, sd, pre + ".cv1", 1, 1, 0)
    for j in range(n):
        t = conv_bn_silu(a, sd, f"{pre}.m.{j}.cv1", 1, 1, 0)
        t = conv_bn_silu(t, sd, f"{pre}.m.{j}.cv2", 3, 1, 1)
        a = a + t if shortcut else t
    b = conv_bn_silu(x, sd, pre + ".cv2", 1, 1, 0)
    return conv_bn_silu(torch.cat((a, b), 1), sd, pre + ".cv3", 1, 1, 0)


def sppf(x, sd, pre, k):
    """models/common.py:252-267: three chained k x k stride-1 max pools, -inf padded."""
    x = conv_bn_silu(x, sd, pre + ".cv1", 1, 1, 0)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv_bn_silu(torch.cat((x, y1, y2, y3), 1), sd, pre + ".cv2", 1, 1, 0)


def adaptive_window(n_in, n_out):
    """models/common.py:879-882: stride = in // out, kernel = in - (out-1)*stride; identity when in <= out."""
    if n_in > n_out:
        s = n_in // n_out
        return n_in - (n_out - 1) * s, s
    return 1, 1


def pooled_tokens(x, va, ha, w1, w2, pos):
    """models/common.py:817-823: w1*avgpool + w2*maxpool, flattened to (B, N, C), plus positional embedding."""
    _, _, h, w = x.shape
    if h > va or w > ha:
        kh, sh = (h - (va - 1) * (h // va)), h // va
        kw, sw = (w - (ha - 1) * (w // ha)), w // ha
        avg = F.avg_pool2d(x, (kh, kw), (sh, sw))
        mx = F.max_pool2d(x, (kh, kw), (sh, sw))
    else:
        avg = mx = x
    t = avg * w1 + mx * w2
    b, c = t.shape[:2]
    return t.reshape(b, c, -1).permute(0, 2, 1) + pos


def layer_norm(x, w, b):
    if x.dtype not in (torch.float32, torch.float64):              # 16-bit mode (OracleModel dtype=): what nn.LayerNorm itself does for a .half() model
        return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def linear(x, sd, pre):
    return x @ sd[pre + ".weight"].t() + sd[pre + ".bias"]


def gelu_erf(x):
    if x.dtype not in (torch.float32, torch.float64):
        return F.gelu(x)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def cross_attention(v, i, sd, pre, heads, store=None):
    """models/common.py:641-687.  Note the crossing: the IR queries attend to the RGB keys/values to produce
    out_vis (:670,:682) and the RGB queries attend to the IR keys/values to produce out_ir (:671,:684).
    store (optional): applied wherever a 16-bit implementation writes a tensor to memory — the normalised tokens, q / k / v, the
    attention output (see cross_transformer)."""
    st = store or (lambda t: t)
    bsz, n, c = v.shape
    dk = c // heads
    vn = st(layer_norm(v, sd[pre + ".LN1.weight"], sd[pre + ".LN1.bias"]))
    inn = st(layer_norm(i, sd[pre + ".LN2.weight"], sd[pre + ".LN2.bias"]))

    def split(t):
        return t.reshape(bsz, n, heads, dk).permute(0, 2, 1, 3)            # (B, h, N, dk)

    q_v, k_v, v_v = (split(st(linear(vn, sd, f"{pre}.{nm}_proj_vis"))) for nm in ("que", "key", "val"))
    q_i, k_i, v_i = (split(st(linear(inn, sd, f"{pre}.{nm}_proj_ir"))) for nm in ("que", "key", "val"))
    scale = 1.0 / math.sqrt(dk)
    a_v = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", q_i, k_v) * scale, -1)
    a_i = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", q_v, k_i) * scale, -1)
    o_v = st(torch.einsum("bhqk,bhkd->bhqd", a_v, v_v).permute(0, 2, 1, 3).reshape(bsz, n, c))
    o_i = st(torch.einsum("bhqk,bhkd->bhqd", a_i, v_i).permute(0, 2, 1, 3).reshape(bsz, n, c))
    return linear(o_v, sd, pre + ".out_proj_vis"), linear(o_i, sd, pre + ".out_proj_ir")


def cross_transformer(v, i, sd, pre, heads, loops, store=None, res32=False):
    """models/common.py:737-759: parameter-shared loop; ONE LN2 normalises both modalities before their MLPs.
    store (optional; tests/test_gpu_dmff_fused.py): a function applied at every point where a 16-bit implementation STORES a tensor
    (normalised tokens, q / k / v, attention output, x_att, the MLP's normalised input, the hidden activations, the block's output).
    With float64 tensors / parameters and store = round-trip through bf16 / f16 this is the reference's arithmetic evaluated exactly,
    with the storage roundings of the HIP kernels and nothing else: what remains between it and a kernel is the kernel's own
    arithmetic error (fp32 accumulation order, exp2 / erf approximations, the rounding of the attention probabilities).
    res32 (with store): the storage pattern of the HIP kernels for loops > 1 since round 5 — the residual chain x -> x_att -> x' is carried
    UNROUNDED from one iteration to the next (an fp32 token stream beside the 16-bit one); attention still reads the stored (rounded) tokens."""
    st = store or (lambda t: t)
    co = [sd[f"{pre}.coefficient{k}.bias"] for k in range(1, 9)]
    ln_w, ln_b = sd[pre + ".LN2.weight"], sd[pre + ".LN2.bias"]
    keep = (lambda t: t) if res32 else st          # what happens to a tensor of the residual chain
    vs, is_ = v, i                                 # the stored (rounded) tokens attention reads
    for _ in range(loops):
        o_v, o_i = cross_attention(vs, is_, sd, pre + ".crossatt", heads, store)
        va = keep(co[0] * v + co[1] * o_v)
        ia = keep(co[2] * i + co[3] * o_i)
        hv = linear(st(gelu_erf(linear(st(layer_norm(va, ln_w, ln_b)), sd, pre + ".mlp_vis.0"))), sd, pre + ".mlp_vis.2")
        hi = linear(st(gelu_erf(linear(st(layer_norm(ia, ln_w, ln_b)), sd, pre + ".mlp_ir.0"))), sd, pre + ".mlp_ir.2")
        v = keep(co[4] * va + co[5] * hv)
        i = keep(co[6] * ia + co[7] * hi)
        vs, is_ = st(v), st(i)
    return vs, is_


def bilinear_resize(t, out_h, out_w):
    """F.interpolate(mode='bilinear', align_corners=False) written out (models/common.py:831,837).
    src = (dst + 0.5) * in/out - 0.5 clamped at 0; neighbours clamped at the border."""
    _, _, h, w = t.shape
    if t.dtype != torch.float32:
        return F.interpolate(t, size=(out_h, out_w), mode="bilinear", align_corners=False)

    def axis(n_in, n_out):
        d = torch.arange(n_out, dtype=torch.float32)
        src = ((d + 0.5) * (float(n_in) / float(n_out)) - 0.5).clamp_min(0.0)
        i0 = src.floor().long().clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        lam = src - i0.float()
        return i0, i1, lam

    y0, y1, ly = axis(h, out_h)
    x0, x1, lx = axis(w, out_w)
    top = t[:, :, y0][:, :, :, x0] * (1 - lx) + t[:, :, y0][:, :, :, x1] * lx
    bot = t[:, :, y1][:, :, :, x0] * (1 - lx) + t[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly)[None, None, :, None] + bot * ly[None, None, :, None]


def dmff(rgb, ir, sd, pre, va, ha, heads, loops):
    """TransformerFusionBlock.forward in eval mode, models/common.py:809-865."""
    b, c, h, w = rgb.shape
    tv = pooled_tokens(rgb, va, ha, sd[pre + ".vis_coefficient.w1"], sd[pre + ".vis_coefficient.w2"],
                       sd[pre + ".pos_emb_vis"])
    ti = pooled_tokens(ir, va, ha, sd[pre + ".ir_coefficient.w1"], sd[pre + ".ir_coefficient.w2"],
                       sd[pre + ".pos_emb_ir"])
    th, tw = (va, ha) if (h > va or w > ha) else (h, w)
    tv, ti = cross_transformer(tv, ti, sd, pre + ".crosstransformer.0", heads, loops)
    fv = bilinear_resize(tv.reshape(b, th, tw, c).permute(0, 3, 1, 2), h, w) + rgb
    fi = bilinear_resize(ti.reshape(b, th, tw, c).permute(0, 3, 1, 2), h, w) + ir
    return conv_bn_silu(torch.cat((fv, fi), 1), sd, pre + ".conv1x1_out", 1, 1, 0)


def detect(feats, sd, pre, nc, anchors):
    """models/yolo_test.py:43-65 (eval branch)."""
    na, no = len(anchors[0]) // 2, nc + 5
    z, logits, raws = [], [], []
    for l, x in enumerate(feats):
        y = F.conv2d(x, sd[f"{pre}.m.{l}.weight"], sd[f"{pre}.m.{l}.bias"])
        b, _, ny, nx = y.shape
        y = y.reshape(b, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        raws.append(y)
        s = torch.sigmoid(y)
        gy, gx = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32),
                                indexing="ij")
        grid = torch.stack((gx, gy), -1)[None, None]
        # 16-bit mode: `model.half()` casts the anchor_grid buffer too, the grid stays fp32 (models/yolo_test.py:69-70), and the
        # decoded values are assigned back INTO the 16-bit tensor y (:57-60), i.e. rounded to the storage type
        anc = torch.tensor(anchors[l], dtype=torch.float32).to(y.dtype).reshape(1, na, 1, 1, 2)
        xy = ((s[..., 0:2] * 2.0 - 0.5 + grid) * STRIDES[l]).to(y.dtype)
        wh = (s[..., 2:4] * 2.0) ** 2 * anc
        z.append(torch.cat((xy, wh, s[..., 4:]), -1).reshape(b, -1, no))
        logits.append(y[..., 5:].reshape(b, -1, nc))
    return torch.cat(z, 1), torch.cat(logits, 1), raws


class OracleModel:
    """Functional interpreter of a *_Transfusion_* yaml (models/yolo_test.py:136-163 graph walk)."""

    def __init__(self, cfg, state_dict, loops=None, dtype=torch.float32):
        """dtype = torch.float16 / bfloat16 restates what the reference does with `model.half()` (detect_twostream.py:40,
        test.py:73-75): every parameter AND buffer in the 16-bit type, every op evaluated by torch in that type (its CPU
        kernels accumulate in fp32 and round each op's output).  Used to measure the reference's own 16-bit deviation from
        its fp32 result on given weights / inputs — the yardstick for the HIP path's 16-bit tolerances."""
        self.cfg = cfg
        self.dtype = dtype
        self.sd = {k: (v.detach().float().to(dtype) if v.is_floating_point() else v.detach()) for k, v in state_dict.items()}
        self.layers, self.ch = build_layers(cfg)
        self.loops = loops

    @torch.no_grad()
    def forward(self, rgb, ir, keep_layers=False):
        sd, outs = self.sd, []
        rgb, ir = rgb.to(self.dtype), ir.to(self.dtype)
        x = rgb
        for (i, f, kind, p) in self.layers:
            pre = f"model.{i}"
            if f == -4:
                src = ir                                             # models/yolo_test.py:154-155
            elif isinstance(f, int):
                src = x if f == -1 else outs[f]
            else:
                src = [x if j == -1 else outs[j] for j in f]
            if kind == "Conv":
                x = conv_bn_silu(src, sd, pre, p["k"], p["s"], p["p"])
            elif kind == "C3":
                x = c3(src, sd, pre, p["n"], p["shortcut"])
            elif kind == "SPPF":
                x = sppf(src, sd, pre, p["k"])
            elif kind == "nn.Upsample":
                x = src.repeat_interleave(int(p["scale"]), 2).repeat_interleave(int(p["scale"]), 3)
            elif kind == "Concat":
                x = torch.cat(src, p["dim"])
            elif kind == "TransformerFusionBlock":
                x = dmff(src[0], src[1], sd, pre, p["va"], p["ha"], p["heads"],
                         self.loops if self.loops is not None else p["loops"])
            elif kind == "NiNfusion":                                 # models/common.py:348-360: SiLU(conv(cat)), no BN, no bias
                y = torch.cat(src, 1)
                x = F.silu(F.conv2d(y, sd[pre + ".conv.weight"], None, p["s"], p["k"] // 2))
            elif kind == "Add":                                       # models/common.py:324-331
                x = src[0] * p["w"] + src[1] * (1 - p["w"])
            elif kind == "Detect":
                x = detect(src, sd, pre, p["nc"], p["anchors"])
            outs.append(x)
        return (x, outs) if keep_layers else x

    __call__ = forward


# --------------------------------------------------------------------------------------------------------------
# NMS                                                             (restates utils/general.py:518-607, :332-339)
# --------------------------------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_NMS_LIB = None


def _nms_lib():
    global _NMS_LIB
    if _NMS_LIB is None:
        path = os.path.join(_HERE, "libnms_oracle.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.nms_greedy_f32.restype = ctypes.c_int
            lib.nms_greedy_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                           ctypes.c_void_p]
            _NMS_LIB = lib
        else:
            _NMS_LIB = False
    return _NMS_LIB


def nms_greedy(boxes, scores, iou_thres):
    """torchvision.ops.nms semantics (third-party, torchvision>=0.8.1 per requirements.txt:11, not vendored):
    visit boxes by descending score (ties: lower index first — stable sort); keep a box unless its IoU with an
    already-kept box is > iou_thres; return kept indices in visiting order.  All arithmetic in fp32, IoU =
    inter / (area_a + area_b - inter)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores, kind="stable").astype(np.int32)
    lib = _nms_lib()
    if lib:
        keep = np.empty(n, np.int32)
        cnt = lib.nms_greedy_f32(boxes.ctypes.data, order.ctypes.data, n, np.float32(iou_thres), keep.ctypes.data)
        return keep[:cnt].astype(np.int64)
    x1, y1, x2, y2 = boxes.T
    area = (x2 - x1) * (y2 - y1)
    dead = np.zeros(n, bool)
    keep = []
    for pos in range(n):
        a = order[pos]
        if dead[a]:
            continue
        keep.append(a)
        rest = order[pos + 1:]
        iw = np.maximum(np.float32(0), np.minimum(x2[a], x2[rest]) - np.maximum(x1[a], x1[rest]))
        ih = np.maximum(np.float32(0), np.minimum(y2[a], y2[rest]) - np.maximum(y1[a], y1[rest]))
        inter = iw * ih
        with np.errstate(invalid="ignore", divide="ignore"):
            iou = inter / (area[a] + area[rest] - inter)
        dead[rest[iou > np.float32(iou_thres)]] = True
    return np.asarray(keep, np.int64)


def non_max_suppression(pred, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        max_det=300, max_nms=30000, max_wh=4096.0, return_indices=False):
    """Per-image candidate filtering + class-offset batched NMS, utils/general.py:518-607 (labels=() and
    merge=False paths; the 10 s wall-clock break at :603-605 is deliberately not restated — SURVEY §5).

    pred: (B, rows, 5+nc) fp32 [cx, cy, w, h, obj, cls...].  Returns list of (n, 6) [x1,y1,x2,y2,conf,cls]."""
    pred = np.asarray(pred, dtype=np.float32)
    nc = pred.shape[2] - 5
    multi_label = multi_label and nc > 1
    outs, idxs = [], []
    for x in pred:
        x = x[x[:, 4] > np.float32(conf_thres)].copy()
        if x.shape[0] == 0:
            outs.append(np.zeros((0, 6), np.float32)); idxs.append(np.zeros((0,), np.int64)); continue
        x[:, 5:] *= x[:, 4:5]
        half = x[:, 2:4] / np.float32(2)
        box = np.concatenate((x[:, 0:2] - half, x[:, 0:2] + half), 1)
        if multi_label:
            i, j = np.nonzero(x[:, 5:] > np.float32(conf_thres))
            det = np.concatenate((box[i], x[i, j + 5][:, None], j[:, None].astype(np.float32)), 1)
        else:
            j = x[:, 5:].argmax(1)
            conf = x[np.arange(x.shape[0]), j + 5]
            det = np.concatenate((box, conf[:, None], j[:, None].astype(np.float32)), 1)[conf > np.float32(conf_thres)]
        if classes is not None:
            det = det[np.isin(det[:, 5], np.asarray(classes, np.float32))]
        n = det.shape[0]
        if n == 0:
            outs.append(np.zeros((0, 6), np.float32)); idxs.append(np.zeros((0,), np.int64)); continue
        if n > max_nms:
            det = det[np.argsort(-det[:, 4], kind="stable")[:max_nms]]
        off = det[:, 5:6] * np.float32(0.0 if agnostic else max_wh)
        keep = nms_greedy(det[:, :4] + off, det[:, 4], iou_thres)[:max_det]
        outs.append(det[keep]); idxs.append(keep)
    return (outs, idxs) if return_indices else outs


# --------------------------------------------------------------------------------------------------------------
# mAP                                                         (restates utils/metrics.py:18-110, test.py:196-230)
# --------------------------------------------------------------------------------------------------------------
def box_iou(a, b):
    """utils/general.py:455-477."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    inter = np.clip(rb - lt, 0, None).prod(2)
    return inter / (area_a[:, None] + area_b[None] - inter)


def match_predictions(det, gt, iouv):
    """test.py:196-230: per class, greedily assign each GT to the best not-yet-used prediction above iouv[0]."""
    correct = np.zeros((det.shape[0], len(iouv)), bool)
    if det.shape[0] == 0 or gt.shape[0] == 0:
        return correct
    found = 0
    for cls in np.unique(gt[:, 0]):
        ti = np.nonzero(gt[:, 0] == cls)[0]
        pi = np.nonzero(det[:, 5] == cls)[0]
        if pi.size == 0:
            continue
        iou = box_iou(det[pi, :4], gt[ti, 1:5])
        best, arg = iou.max(1), iou.argmax(1)
        used = set()                        # per class (test.py:219); the found-count below spans classes (:224)
        for j in np.nonzero(best > iouv[0])[0]:
            d = ti[arg[j]]
            if d not in used:
                used.add(d)
                found += 1
                correct[pi[j]] = best[j] > iouv
                if found == gt.shape[0]:
                    break
    return correct


def average_precision(recall, precision):
    """utils/metrics.py:85-110 ('interp' method: 101-point integration of the precision envelope)."""
    mrec = np.concatenate(([0.0], recall, [recall[-1] + 0.01]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    return np.trapezoid(np.interp(x, mrec, mpre), x)


def ap_per_class(tp, conf, pred_cls, target_cls):
    """utils/metrics.py:18-82 without plotting: returns ap (n_cls, n_iou), classes."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    ap = np.zeros((classes.shape[0], tp.shape[1]))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = int((target_cls == c).sum()), int(sel.sum())
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + 1e-16)
        precision = tpc / (tpc + fpc)
        for j in range(tp.shape[1]):
            ap[ci, j] = average_precision(recall[:, j], precision[:, j])
    return ap, classes.astype(np.int32)
