"""Module library of the two-stream detector, MI355X edition.

Same class names, constructor signatures, attribute names and state_dict keys as the reference's
models/common.py (so `parse_model`'s name lookup, pickled checkpoints and `load_state_dict(strict=True)` keep
working — SURVEY.md §8b), but the modules do no arithmetic themselves: each one *emits* HIP kernel launches into
an `engine.Plan` over NHWC buffers.  `forward(x)` on a single module builds and runs a one-module plan, so the
classes still duck-type as nn.Modules taking/returning NCHW-shaped tensors.

There is deliberately no PyTorch / CPU fallback: calling a module with CPU tensors, in training mode, or without
libicaf.so raises.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..options import OPT
from ..engine import ImageIn, Plan, concat_view, from_act, to_act

BN_EPS_DEFAULT = 1e-5


def autopad(k, p=None):
    """'same' padding for odd kernels (reference models/common.py:36-40)."""
    if p is None:
        p = k // 2 if isinstance(k, int) else [v // 2 for v in k]
    return p


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _module_dtype(m, fallback=torch.float32):
    forced = getattr(m, "compute_dtype", None)
    if forced is not None:
        return forced
    for p in m.parameters():
        return p.dtype
    return fallback


class HipModule(nn.Module):
    """Base: stand-alone execution of `emit` + packed-weight cache."""

    def invalidate(self):
        for m in self.modules():
            if hasattr(m, "_cache"):
                m._cache = {}
            if hasattr(m, "_plans"):
                m._plans = {}

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self.__dict__["_cache"] = {}
        self.__dict__["_plans"] = {}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.__dict__["_cache"] = {}
        self.__dict__["_plans"] = {}
        return r

    def _cached(self, key, make):
        c = self.__dict__.setdefault("_cache", {})
        if key not in c:
            c[key] = make()
        return c[key]

    # stand-alone call: NCHW tensor(s) in, NCHW-shaped tensor(s) out
    def forward(self, x):
        xs = x if isinstance(x, (list, tuple)) else [x]
        if self.training:
            raise NotImplementedError("icafusion_amd implements the eval-mode inference path only (call .eval())")
        for t in xs:
            if not t.is_cuda:
                raise RuntimeError("icafusion_amd modules run on the MI355X only: move inputs to cuda "
                                   "(there is no CPU fallback; the CPU reference lives in oracle/)")
        dt = _module_dtype(self, xs[0].dtype)
        key = (tuple(tuple(t.shape) for t in xs), dt, xs[0].device)
        plans = self.__dict__.setdefault("_plans", {})
        if key not in plans:
            plan = Plan(xs[0].device, dt)
            ins = []
            for t in xs:
                if t.shape[1] % ops.VEC[dt] != 0:
                    buf = torch.zeros(t.shape, dtype=torch.float32, device=t.device)
                    ins.append(ImageIn(buf))
                else:
                    ins.append(plan.act(t.shape[0], t.shape[2], t.shape[3], t.shape[1]))
            plan.inputs = ins
            plan.outputs = self.emit(plan, ins if isinstance(x, (list, tuple)) else ins[0])
            plans[key] = plan
        plan = plans[key]
        for src, dst in zip(xs, plan.inputs):
            if isinstance(dst, ImageIn):
                dst.t.copy_(src)
            else:
                dst.copy_(src.permute(0, 2, 3, 1))
        plan.run()
        out = plan.outputs
        if isinstance(out, torch.Tensor) and out.dim() == 4:
            return from_act(out).clone(memory_format=torch.preserve_format)
        return out


# ----------------------------------------------------------------------------------------------------------
# convolution family
# ----------------------------------------------------------------------------------------------------------
class Conv(HipModule):
    """Conv2d(bias=False) + BatchNorm2d + SiLU (reference models/common.py:48-60).  On the device this is one
    implicit-GEMM launch with BN folded into the packed weights (as utils/torch_utils.py:182-202 does offline) and
    bias + SiLU in the epilogue; `fuse()`d modules (no .bn) are handled identically."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    fuse_stem = True     # image-fed 6x6/s2 layers: staging + convolution as one persistent kernel (16-bit types)

    def fuseforward(self, x):           # kept for API parity with Model.fuse(); same device path
        return self.forward(x)

    def _act_code(self):
        if isinstance(self.act, nn.SiLU):
            return ops.ACT_SILU
        if isinstance(self.act, nn.Identity):
            return ops.ACT_NONE
        if isinstance(self.act, nn.GELU):
            return ops.ACT_GELU
        raise NotImplementedError(f"activation {type(self.act).__name__} is outside the hot path")

    def folded(self):
        """fp32 (weight, bias) with BatchNorm folded in."""
        w = self.conv.weight.detach().float()
        b = self.conv.bias.detach().float() if self.conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        if hasattr(self, "bn"):
            bn = self.bn
            scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            w = w * scale[:, None, None, None]
            b = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
        return w, b

    def emit(self, plan, x, out=None, res=None, twin=None, also=(), twin_also=(), pre_term=None, swap_halves=False,
             chain=None, pre_nearest=False, cin_slice=None):
        """Append this layer's launch.

        twin: the structurally identical Conv of the other backbone stream — x / out / res are then pair acts
              (2, B, H, W, C) and both streams run as ONE groups=2 launch with per-stream weights.
        also: further Convs with the same geometry and activation reading the same input (C3's cv1 and cv2): their
              output channels are appended to this layer's, one GEMM with N = sum of the widths (twin_also: the
              twin stream's counterparts).
        pre_term: fp32 coarse map (B, h, w, Cout) added, bilinearly resized (pre_nearest: nearest), before bias + activation
              (icaf.h).
        cin_slice: (k0, k1) — use only these input-channel columns of the weights (x then has k1 - k0 channels): the
              full-resolution half of a 1x1 conv over cat(up(a), b), whose other half arrives as the nearest pre_term.
        chain: (convs, twin_convs, y2[, keep[, x2]]) — 1x1 SiLU Convs (their outputs concatenated, e.g. the cv1 | cv2 of the C3
              behind a down-sampling Conv) applied to this layer's output tile inside the same launch; only y2 is written —
              unless keep is set: then `out` (residual included) is written too and the 1x1 consumes it as stored (a
              Bottleneck's 3x3 followed by the next Bottleneck's 1x1).  x2: the chained Conv reads cat(this layer's output, x2)
              (a C3's last Bottleneck carrying cv3, x2 = cv2's output: C3.emit).
        swap_halves: the input view holds the two halves of the layer's input channels in swapped order (C3 after an
              odd number of fused Bottlenecks): the weight columns are swapped to match when they are packed."""
        if self.conv.groups != 1 or self.conv.dilation != (1, 1):
            raise NotImplementedError("grouped / dilated convolutions are outside the hot path")
        kh, kw = self.conv.kernel_size
        sh, sw = self.conv.stride
        ph, pw = _pair(self.conv.padding)
        c1 = self.conv.in_channels
        c2 = self.conv.out_channels + sum(e.conv.out_channels for e in also)
        if cin_slice is not None:
            assert (kh, kw) == (1, 1) and not swap_halves and 0 <= cin_slice[0] < cin_slice[1] <= c1
        for e in also:
            assert (e.conv.kernel_size, e.conv.stride, _pair(e.conv.padding), e.conv.in_channels, e._act_code()) == \
                   ((kh, kw), (sh, sw), (ph, pw), c1, self._act_code()), "fused convs must share geometry"
        vec = ops.VEC[plan.dtype]
        paired = twin is not None
        assert len(twin_also) == (len(also) if paired else 0)

        def streams():
            """[(Conv, ...)] per stream: the convs whose folded weights are concatenated along Cout."""
            rows = [(self,) + tuple(also)]
            if paired:
                rows.append((twin,) + tuple(twin_also))
            return rows
        key_tail = (plan.dtype, plan.device, id(twin), tuple(id(e) for e in also), bool(swap_halves))
        if cin_slice is not None:
            key_tail += (tuple(cin_slice),)

        def pack(transform, cin_pad):
            packs = []
            for convs in streams():
                ws, bs = zip(*(c.folded() for c in convs))
                w, b = torch.cat(ws), torch.cat(bs)
                if swap_halves:
                    h = w.shape[1] // 2
                    w = torch.cat((w[:, h:], w[:, :h]), 1)
                if cin_slice is not None:
                    w = w[:, cin_slice[0]:cin_slice[1]]
                wp, kp = ops.pack_conv_weight(transform(w), plan.dtype, cin_pad)
                packs.append((wp, kp, ops.pack_bias(b, c2)))
            if not paired:
                return packs[0]
            return (torch.stack([p[0] for p in packs]).contiguous(), packs[0][1],
                    torch.stack([p[2] for p in packs]).contiguous())

        if isinstance(x, ImageIn):
            assert x.pair == paired
            B, _, H, W = x.shape
            s2d = (kh, kw, sh, sw, ph, pw) == (6, 6, 2, 2, 2, 2) and H % 2 == 0 and W % 2 == 0
            if (s2d and self.fuse_stem and plan.dtype in (torch.bfloat16, torch.float16) and c1 == 3 and c2 in (32, 64)
                    and not also and res is None and (not x.u8 or (paired and x.c0 == 0))):
                # staging + convolution in one persistent kernel reading the NCHW image itself (stem.hip)
                wp, kp, bp = self._cached(("s2d",) + key_tail, lambda: pack(ops.s2d_conv_weight, 16))
                if out is None:
                    out = plan.act(B, H // 2, W // 2, c2, pair=paired)
                plan.add(ops.stem(x.t, wp, kp, bp, out, c2))
                return out
            if s2d:                       # 6x6/s2/p2 over the image == 3x3/s1/p1 over space-to-depth(image)
                cpad = -(-4 * c1 // vec) * vec
                pre = plan.act(B, H // 2, W // 2, cpad, pair=paired)
                plan.add(ops.preprocess_u8(x.t, pre, 1, x.c0, name="preprocess_u8_s2d") if x.u8
                         else ops.preprocess(x.t, pre, 1, name="preprocess_s2d"))
                wp, kp, bp = self._cached(("s2d",) + key_tail, lambda: pack(ops.s2d_conv_weight, cpad))
                x, c1, (kh, kw, sh, sw, ph, pw) = pre, cpad, (3, 3, 1, 1, 1, 1)
            else:
                cpad = -(-c1 // vec) * vec
                pre = plan.act(B, H, W, cpad, pair=paired)
                plan.add(ops.preprocess_u8(x.t, pre, 0, x.c0, name="preprocess_u8_pad") if x.u8
                         else ops.preprocess(x.t, pre, 0, name="preprocess_pad"))
                wp, kp, bp = self._cached(("pad",) + key_tail, lambda: pack(lambda w: w, cpad))
                x, c1 = pre, cpad
        else:
            assert (x.dim() == 5) == paired
            if cin_slice is not None:
                c1 = cin_slice[1] - cin_slice[0]
            if x.shape[-1] != c1:
                raise ValueError(f"Conv expects {c1} input channels, got {x.shape[-1]}")
            if c1 % vec:
                raise NotImplementedError(f"channel count {c1} must be a multiple of {vec} for dtype {plan.dtype}")
            wp, kp, bp = self._cached(("std",) + key_tail, lambda: pack(lambda w: w, None))
        B, H, W = x.shape[-4:-1]
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        if out is None:
            out = plan.act(B, Ho, Wo, c2, pair=paired)
        ch = None
        if chain is not None:
            convs, twin_convs, y2 = chain[:3]
            keep = len(chain) > 3 and bool(chain[3])
            x2 = chain[4] if len(chain) > 4 else None
            n2 = sum(c.conv.out_channels for c in convs)

            def pack2():
                packs = []
                for row in ([convs] + ([twin_convs] if paired else [])):
                    ws, bs = zip(*(c.folded() for c in row))
                    w2p, kp2 = ops.pack_conv_weight(torch.cat(ws), plan.dtype)
                    packs.append((w2p, kp2, ops.pack_bias(torch.cat(bs), n2)))
                if not paired:
                    return packs[0]
                return (torch.stack([p[0] for p in packs]).contiguous(), packs[0][1], torch.stack([p[2] for p in packs]).contiguous())
            w2p, kp2, b2p = self._cached(("chain",) + key_tail + tuple(id(c) for c in convs), pack2)
            ch = dict(w=w2p, kp=kp2, bias=b2p, y=y2, cout=n2, keep=keep)
            if x2 is not None:
                ch["x2"] = x2
            if not keep:
                out = y2[..., :c2] if y2.shape[-1] >= c2 else plan.act(B, Ho, Wo, c2, pair=paired)   # (y is ignored by the kernel)
        plan.add(ops.conv2d(x, wp, kp, bp, out, kh, kw, sh, sw, ph, pw, c1, c2, self._act_code(), res=res,
                            name=f"conv{kh}x{kw}s{sh}" + (("+cv3" if ch.get("x2") is not None else "+1x1") if ch else ""), pre=pre_term, chain=ch,
                            pre_nearest=pre_nearest))
        return (out if ch["keep"] else y2) if ch else out

    fuse_stem2 = True    # stem + the 3x3/s2 Conv behind it + that Conv's chained cv1 | cv2 as ONE persistent kernel

    def stem2_ok(self, plan, x, nxt, c3):
        """self = image-fed 6x6/s2 stem, nxt = the Conv behind it, c3 = the C3 behind that: can icaf_stem2 run all three?
        (mirrors its argument checks: 16-bit, 3 -> 32 -> 64 -> 2 x 32 channels, i.e. the yolov5s width)"""
        k0, k1 = self.conv, getattr(nxt, "conv", None)
        if not (self.fuse_stem2 and self.fuse_stem and isinstance(x, ImageIn) and x.pair and isinstance(nxt, Conv)
                and plan.dtype in (torch.bfloat16, torch.float16) and (not x.u8 or x.c0 == 0)):
            return False
        _, _, H, W = x.shape
        return ((k0.kernel_size, k0.stride, _pair(k0.padding), k0.in_channels, k0.out_channels) == ((6, 6), (2, 2), (2, 2), 3, 32)
                and k0.groups == 1 and isinstance(self.act, nn.SiLU) and H % 2 == 0 and W % 2 == 0
                and (k1.kernel_size, k1.stride, _pair(k1.padding), k1.in_channels, k1.out_channels) == ((3, 3), (2, 2), (1, 1), 32, 64)
                and nxt.chain_ok(plan, c3) and 2 * c3.cv1.conv.out_channels == 64)

    def emit_stem2(self, plan, x, twin, nxt, nxt_twin, convs, twin_convs, y2):
        """Rows 0-2a of both streams as one launch; y2 = the pair act the C3's cv1 | cv2 write."""
        key_tail = (plan.dtype, plan.device, id(twin))

        def stack(packs):
            return (torch.stack([p[0] for p in packs]).contiguous(), packs[0][1], torch.stack([p[2] for p in packs]).contiguous())

        def pack_row(rows, transform, cin_pad):
            packs = []
            for convs_ in rows:
                ws, bs = zip(*(c.folded() for c in convs_))
                w, b = torch.cat(ws), torch.cat(bs)
                wp, kp = ops.pack_conv_weight(transform(w), plan.dtype, cin_pad)
                packs.append((wp, kp, ops.pack_bias(b, w.shape[0])))
            return stack(packs)
        w0, kp0, b0 = self._cached(("s2d",) + key_tail + ((), False), lambda: pack_row([(self,), (twin,)], ops.s2d_conv_weight, 16))
        w1, kp1, b1 = nxt._cached(("std",) + (plan.dtype, plan.device, id(nxt_twin), (), False),
                                  lambda: pack_row([(nxt,), (nxt_twin,)], lambda w: w, None))
        w2, kp2, b2 = nxt._cached(("chain",) + (plan.dtype, plan.device, id(nxt_twin), (), False) + tuple(id(c) for c in convs),
                                  lambda: pack_row([convs, twin_convs], lambda w: w, None))
        n2 = sum(c.conv.out_channels for c in convs)
        plan.add(ops.stem2(x.t, w0, kp0, b0, w1, kp1, b1, w2, kp2, b2, y2, self.conv.out_channels, nxt.conv.out_channels, n2))
        return y2

    chain_fuse = True    # let a C3 behind this Conv run its cv1 | cv2 GEMM on this layer's output tile (one launch)

    def chain_ok(self, plan, c3):
        """Can the fused cv1 | cv2 GEMM of `c3` ride on this conv's output tile?  (mirrors the checks of icaf_conv2d)"""
        k, cv = self.conv, c3.cv1.conv
        n1, n2 = k.out_channels, 2 * cv.out_channels
        return (self.chain_fuse and plan.dtype in (torch.bfloat16, torch.float16) and isinstance(self.act, nn.SiLU)
                and hasattr(k, "kernel_size") and k.kernel_size == (3, 3) and k.groups == 1 and k.in_channels % 32 == 0
                and cv.kernel_size == (1, 1) and cv.stride == (1, 1) and cv.in_channels == n1
                and c3.cv2.conv.out_channels == cv.out_channels and isinstance(c3.cv1.act, nn.SiLU)
                and max(n1, n2) <= self.chain_max_width)

    chain_max_width = 128

    def chain_ok_1x1(self, plan, nxt):
        """Can the 1x1 SiLU Conv `nxt` (the next Bottleneck's cv1) ride on this 3x3 layer's output tile, this layer's output
        (shortcut included) being written as well?  (icaf_conv2d: chain_keep)"""
        k, kn = self.conv, nxt.conv
        return (self.chain_fuse and plan.dtype in (torch.bfloat16, torch.float16) and isinstance(self.act, nn.SiLU)
                and isinstance(nxt.act, nn.SiLU) and k.kernel_size == (3, 3) and k.groups == 1 and k.in_channels % 64 == 0
                and kn.kernel_size == (1, 1) and kn.stride == (1, 1) and kn.groups == 1 and kn.in_channels == k.out_channels
                and max(k.out_channels, kn.out_channels) <= self.chain_max_width)

    chain_tail = OPT.c3_tail      # A/B switch: a C3's cv3 rides on its last Bottleneck's 3x3 (a class default, as fuse_decode)

    def chain_ok_tail(self, plan, cv3):
        """Can the C3's cv3 (1x1 SiLU over cat(m, cv2)) ride on this 3x3 layer — the block's last Bottleneck.cv2 — so that neither m nor
        the concatenation reach HBM?  (icaf_conv2d: x2; built in cwide.hip for 128 -> 128 layers with 256 output channels of cv3)"""
        k, k3 = self.conv, cv3.conv
        return (self.chain_tail and OPT.cwide and plan.dtype in (torch.bfloat16, torch.float16) and isinstance(self.act, nn.SiLU)
                and isinstance(cv3.act, nn.SiLU) and (k.kernel_size, k.stride, _pair(k.padding), k.groups) == ((3, 3), (1, 1), (1, 1), 1)
                and (k.in_channels, k.out_channels) == (128, 128) and k3.kernel_size == (1, 1) and k3.stride == (1, 1) and k3.groups == 1
                and _pair(k3.padding) == (0, 0) and k3.in_channels == 2 * k.out_channels and k3.out_channels == 256)


class Bottleneck(HipModule):
    """1x1 -> 3x3 with optional identity shortcut (reference models/common.py:184-194); the shortcut add is the
    residual term of the second conv's epilogue (which may write in place over its own residual: every output
    element is read and written by the same thread, and the GEMM's input is the 1x1's separate output)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def emit(self, plan, x, out=None, twin=None):
        t = self.cv1.emit(plan, x, twin=twin.cv1 if twin is not None else None)
        return self.cv2.emit(plan, t, out=out, res=x if self.add else None,
                             twin=twin.cv2 if twin is not None else None)

    fuse = True      # use the one-launch kernel (icaf_bottleneck) where it is built and measured faster
    fuse_widths = (32,)

    def fusable(self, plan, x):
        """One-launch form: built for c_ in {32, 64} (16-bit types, 1x1 then 3x3 / s1 / p1, SiLU); USED for c_ = 32 on wide
        maps only: measured on MI355X (both streams, batch 32) 174 us against 73 + 112 us at 160x160 / c_ = 32, but
        155 us against 30 + 62 us at 80x80 / c_ = 64, where two LDS patches leave room for one workgroup per CU."""
        a, b = self.cv1.conv, self.cv2.conv
        c = a.in_channels
        return (self.fuse and plan.dtype in (torch.bfloat16, torch.float16) and c in self.fuse_widths
                and a.out_channels == c and b.in_channels == c and b.out_channels == c
                and a.kernel_size == (1, 1) and a.stride == (1, 1) and _pair(a.padding) == (0, 0)
                and b.kernel_size == (3, 3) and b.stride == (1, 1) and _pair(b.padding) == (1, 1)
                and a.groups == 1 and b.groups == 1 and isinstance(self.cv1.act, nn.SiLU) and isinstance(self.cv2.act, nn.SiLU)
                and x.shape[-2] >= 64)

    def emit_fused(self, plan, x, out, twin=None, cv3=None):
        """y = [x +] SiLU(conv3x3(SiLU(conv1x1(x)))) as ONE launch; `out` must be a different buffer (slice) than x.
        cv3 = (Conv, twin Conv or None, x2, y3): the C3's cv3 rides on the launch — x2 is the cv2 half of its input, y3 its
        output; the Bottleneck's own output is then never written (`out` is ignored)."""
        c = self.cv1.conv.in_channels
        paired = twin is not None
        mods = [self] + ([twin] if paired else [])
        tail = None
        if cv3 is not None:
            k3, k3t, x2, y3 = cv3

            def pack3():          # K columns in the order [cv2 | m]: exactly Conv.emit(swap_halves=True)'s packing (shared cache key)
                packs = []
                for cv in [k3] + ([k3t] if paired else []):
                    w, b = cv.folded()
                    h = w.shape[1] // 2
                    wp, kp = ops.pack_conv_weight(torch.cat((w[:, h:], w[:, :h]), 1), plan.dtype)
                    packs.append((wp, kp, ops.pack_bias(b, w.shape[0])))
                if not paired:
                    return packs[0]
                return (torch.stack([p[0] for p in packs]).contiguous(), packs[0][1], torch.stack([p[2] for p in packs]).contiguous())
            w3, kp3, b3 = k3._cached(("std", plan.dtype, plan.device, id(k3t), (), True), pack3)
            tail = dict(w=w3, kp=kp3, bias=b3, y=y3, cout=k3.conv.out_channels, x2=x2)

        def make():
            p1 = [ops.pack_conv_weight(m.cv1.folded()[0], plan.dtype) for m in mods]
            p2 = [ops.pack_conv_weight(m.cv2.folded()[0], plan.dtype) for m in mods]
            b1 = [ops.pack_bias(m.cv1.folded()[1], c) for m in mods]
            b2 = [ops.pack_bias(m.cv2.folded()[1], c) for m in mods]
            st = (lambda ts: torch.stack(ts).contiguous()) if paired else (lambda ts: ts[0])
            return st([p[0] for p in p1]), p1[0][1], st(b1), st([p[0] for p in p2]), p2[0][1], st(b2)
        w1, kp1, b1, w2, kp2, b2 = self._cached(("bneck", plan.dtype, plan.device, id(twin)), make)
        plan.add(ops.bottleneck(x, w1, kp1, b1, w2, kp2, b2, None if tail else out, c, self.add, 1 if c == 32 else 2, cv3=tail))
        return tail["y"] if tail else out


class C3(HipModule):
    """CSP bottleneck with three convs (reference models/common.py:216-227).  cv1 and cv2 read the same input, so they
    run as ONE GEMM writing both halves of the buffer cv3 reads (the torch.cat is never materialised); the
    bottleneck chain then updates the first half in place."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    fuse_cv3 = True      # n = 1, c_ = 32: Bottleneck + cv3 as one launch (icaf_bottleneck with a chained cv3)
    chain_bottlenecks = True     # n > 1: a Bottleneck's 3x3 (+ shortcut) and the next Bottleneck's 1x1 as one launch

    def emit(self, plan, x, out=None, twin=None, lead=None):
        """lead = (Conv, twin Conv or None): the down-sampling Conv in front of this block whose output only this block
        reads; `x` is then THAT conv's input and cv1 | cv2 run chained on its output tile (the tensor between the two
        yaml rows is never written)."""
        vcat = x if isinstance(x, VirtualCat) else None
        if vcat is not None:
            assert lead is None and twin is None
            x = vcat.other
        if lead is not None and len(lead) == 4:        # (Conv, twin, stem, stem twin): x is the image pair itself
            B, _, H, W = x.shape
            H, W = (H // 2 - 1) // 2 + 1, (W // 2 - 1) // 2 + 1
        elif lead is not None:
            k = lead[0].conv
            B = x.shape[-4]
            H = (x.shape[-3] + 2 * k.padding[0] - k.kernel_size[0]) // k.stride[0] + 1
            W = (x.shape[-2] + 2 * k.padding[1] - k.kernel_size[1]) // k.stride[1] + 1
        else:
            B, H, W = x.shape[-4:-1]
        c_ = self.cv1.conv.out_channels
        paired = twin is not None
        fused = [blk.fusable(plan, x if lead is None else torch.empty((B, H, W, 1), device="meta")) for blk in self.m]
        # Buffer of three c_-wide slots [a | b | a']: cv1|cv2 write [a | b]; a fused Bottleneck cannot run in place (its
        # neighbours' patches read x), so the chain ping-pongs between slot 0 and slot 2; cv3 then reads [a | b] or
        # [b | a'] — in the second case with its weight columns swapped to match.
        k3 = self.cv3.conv
        tail3 = (self.fuse_cv3 and len(self.m) == 1 and fused[0] and c_ == 32 and k3.out_channels == 64 and k3.kernel_size == (1, 1)
                 and k3.stride == (1, 1) and k3.groups == 1 and k3.in_channels == 2 * c_ and isinstance(self.cv3.act, nn.SiLU))
        cat = plan.act(B, H, W, (3 if any(fused) and not tail3 else 2) * c_, pair=paired)
        if lead is not None and len(lead) == 4:
            lead[2].emit_stem2(plan, x, lead[3], lead[0], lead[1], (self.cv1, self.cv2), (twin.cv1, twin.cv2), cat[..., :2 * c_])
        elif lead is not None:
            lead[0].emit(plan, x, twin=lead[1], chain=((self.cv1, self.cv2), (twin.cv1, twin.cv2) if paired else None,
                                                       cat[..., :2 * c_]))
        elif vcat is not None:
            # cv(cat(up(a), b)) = SiLU(up(Wa . a) + Wb . b + bias): the 1x1 commutes with the nearest up-sampling, so the
            # a-half of cv1 | cv2 runs at LOW resolution (fp32 out, a quarter of the pixels) and enters the GEMM over b as its
            # nearest-resized pre-activation term; nn.Upsample's output and the Concat buffer are never written or read.
            ca, cb = vcat.low.shape[-1], vcat.other.shape[-1]

            def make_a():
                w = torch.cat([self.cv1.folded()[0], self.cv2.folded()[0]])[:, :ca]
                return ops.pack_conv_weight(w, plan.dtype)
            wa, kpa = self._cached(("upterm", plan.dtype, plan.device, ca), make_a)
            Bl, hl, wl, _ = vcat.low.shape
            P = plan.empty((Bl, hl, wl, 2 * c_), torch.float32)
            plan.add(ops.conv2d(vcat.low, wa, kpa, None, P, 1, 1, 1, 1, 0, 0, ca, 2 * c_, ops.ACT_NONE, name="c3_up_term"))
            self.cv1.emit(plan, x, out=cat[..., :2 * c_], also=(self.cv2,), pre_term=P, pre_nearest=True, cin_slice=(ca, ca + cb))
        else:
            self.cv1.emit(plan, x, out=cat[..., :2 * c_], twin=twin.cv1 if paired else None, also=(self.cv2,),
                          twin_also=(twin.cv2,) if paired else ())
        if tail3:         # the single fused Bottleneck carries cv3 as well: cat(m, cv2) and m never reach HBM
            if out is None:
                out = plan.act(B, H, W, k3.out_channels, pair=paired)
            return self.m[0].emit_fused(plan, cat[..., :c_], None, twin=twin.m[0] if paired else None,
                                        cv3=(self.cv3, twin.cv3 if paired else None, cat[..., c_:2 * c_], out))
        cur = 0
        t_next = None                       # the next Bottleneck's 1x1 output, when the previous 3x3 launch produced it
        for j, blk in enumerate(self.m):
            a = cat[..., cur * c_:(cur + 1) * c_]
            tw = twin.m[j] if paired else None
            if fused[j]:
                cur = 2 - cur
                blk.emit_fused(plan, a, cat[..., cur * c_:(cur + 1) * c_], twin=tw)
                continue
            nxt = self.m[j + 1] if j + 1 < len(self.m) and not fused[j + 1] else None
            t = t_next if t_next is not None else blk.cv1.emit(plan, a, twin=tw.cv1 if paired else None)
            t_next, ch = None, None
            if nxt is not None and self.chain_bottlenecks and blk.cv2.chain_ok_1x1(plan, nxt.cv1):
                t_next = plan.act(B, H, W, nxt.cv1.conv.out_channels, pair=paired)
                ch = ((nxt.cv1,), (twin.m[j + 1].cv1,) if paired else None, t_next, True)
            if j == len(self.m) - 1 and cur == 0 and blk.cv2.chain_ok_tail(plan, self.cv3):
                # the block's tail as ONE launch: m = [a +] SiLU(conv3x3(t)) stays in the kernel's staging tile, cv3 reads [m | b] from there
                # and from cv2's half of the buffer (icaf.h: icaf_conv_args.x2) — m is never written, cat(m, b) never read
                if out is None:
                    out = plan.act(B, H, W, k3.out_channels, pair=paired)
                return blk.cv2.emit(plan, t, out=a, res=a if blk.add else None, twin=tw.cv2 if paired else None,
                                    chain=((self.cv3,), (twin.cv3,) if paired else None, out, False, cat[..., c_:2 * c_]))
            blk.cv2.emit(plan, t, out=a, res=a if blk.add else None, twin=tw.cv2 if paired else None, chain=ch)
        src = cat[..., :2 * c_] if cur == 0 else cat[..., c_:3 * c_]
        return self.cv3.emit(plan, src, out=out, twin=twin.cv3 if paired else None, swap_halves=cur != 0)


class SPPF(HipModule):
    """Spatial pyramid pooling - fast (reference models/common.py:252-267): the three chained max pools run as
    one kernel writing the three extra channel groups of the buffer cv2 reads."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def emit(self, plan, x, out=None, twin=None):
        B, H, W = x.shape[-4:-1]
        c_ = self.cv1.conv.out_channels
        k = self.m.kernel_size if isinstance(self.m.kernel_size, int) else self.m.kernel_size[0]
        paired = twin is not None
        cat = plan.act(B, H, W, 4 * c_, pair=paired)
        self.cv1.emit(plan, x, out=cat[..., :c_], twin=twin.cv1 if paired else None)
        plan.add(ops.sppf_pool(cat[..., :c_], cat[..., c_:2 * c_], cat[..., 2 * c_:3 * c_], cat[..., 3 * c_:], k))
        return self.cv2.emit(plan, cat, out=out, twin=twin.cv2 if paired else None)


class VirtualCat:
    """cat(nearest_up(low, scale), other) that is never materialised: a C3 consumes it (C3.emit)."""

    def __init__(self, low, scale, other):
        self.low, self.scale, self.other = low, scale, other
        B, h, w, _ = low.shape
        assert other.shape[:3] == (B, h * scale, w * scale)
        self.shape = (B, h * scale, w * scale, low.shape[3] + other.shape[3])


class Concat(HipModule):
    """Channel concatenation (reference models/common.py:313-321).  When the producers already wrote adjacent
    slices of one buffer (the normal case inside Model) this is free."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def emit(self, plan, xs, out=None):
        if self.d != 1:
            raise NotImplementedError("only channel concatenation is on the hot path")
        v = concat_view(xs)
        if v is not None and out is None:
            return v
        B, H, W, _ = xs[0].shape
        if out is None:
            out = plan.act(B, H, W, sum(t.shape[3] for t in xs))
        c0 = 0
        for t in xs:
            plan.add(ops.copy_channels(t, out[..., c0:c0 + t.shape[3]]))
            c0 += t.shape[3]
        return out


class Add(HipModule):
    """Weighted sum of the two streams (reference models/common.py:324-331): x[0]*w + x[1]*(1-w)."""

    def __init__(self, weight=0.5):
        super().__init__()
        self.w = weight

    def emit(self, plan, xs, out=None):
        a, b = xs
        if out is None:
            out = plan.act(*a.shape)
        plan.add(ops.axpby(a, b, out, self.w, 1 - self.w))
        return out


class NiNfusion(HipModule):
    """Concat + k x k convolution (no BN, no bias) + SiLU (reference models/common.py:348-360): one GEMM over the two
    streams' features, which Model.build_plan places as adjacent channel slices so the concat is free."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1):
        super().__init__()
        self.concat = Concat(dimension=1)
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.act = nn.SiLU()

    # the packed-weight / launch logic is Conv's (a Conv without .bn and without bias)
    folded = Conv.folded
    _act_code = Conv._act_code
    fuse_stem = False

    def emit(self, plan, xs, out=None):
        x = self.concat.emit(plan, list(xs))
        return Conv.emit(self, plan, x, out=out)


def emit_upsample(m, plan, x, out=None):
    """nn.Upsample(None, 2, 'nearest') rows of the head (yaml rows 24 / 28)."""
    if m.mode != "nearest" or m.scale_factor is None or float(m.scale_factor) != int(m.scale_factor):
        raise NotImplementedError("only integer-factor nearest upsampling is on the hot path")
    s = int(m.scale_factor)
    B, H, W, C = x.shape
    if out is None:
        out = plan.act(B, H * s, W * s, C)
    plan.add(ops.upsample_nearest(x, out, s))
    return out


# ----------------------------------------------------------------------------------------------------------
# DMFF
# ----------------------------------------------------------------------------------------------------------
class LearnableCoefficient(nn.Module):
    """Scalar gain, init 1 (reference models/common.py:569-576); folded into GEMM epilogue alphas."""

    def __init__(self):
        super().__init__()
        self.bias = nn.Parameter(torch.FloatTensor([1.0]), requires_grad=True)


class LearnableWeights(nn.Module):
    """Two scalar mixing weights, init 0.5 (reference models/common.py:579-587)."""

    def __init__(self):
        super().__init__()
        self.w1 = nn.Parameter(torch.tensor([0.5]), requires_grad=True)
        self.w2 = nn.Parameter(torch.tensor([0.5]), requires_grad=True)


class AdaptivePool2d(nn.Module):
    """Fixed-output pooling with stride = in // out, kernel = in - (out-1)*stride (reference
    models/common.py:868-891).  Holds geometry only; the arithmetic is in icaf_dmff_pool_tokens."""

    def __init__(self, output_h, output_w, pool_type="avg"):
        super().__init__()
        self.output_h, self.output_w, self.pool_type = output_h, output_w, pool_type

    def window(self, h, w):
        """-> (out_h, out_w, kh, kw, sh, sw)"""
        if h > self.output_h or w > self.output_w:
            sh, sw = h // self.output_h, w // self.output_w
            if sh == 0 or sw == 0:
                raise ValueError(f"feature map {h}x{w} too small for {self.output_h}x{self.output_w} anchors")
            return (self.output_h, self.output_w, h - (self.output_h - 1) * sh, w - (self.output_w - 1) * sw, sh, sw)
        return h, w, 1, 1, 1, 1


class CrossAttention(HipModule):
    """Bidirectional cross-modal attention parameters (reference models/common.py:590-687)."""

    def __init__(self, d_model, d_k, d_v, h, attn_pdrop=.1, resid_pdrop=.1):
        super().__init__()
        assert d_k % h == 0
        self.d_model, self.h = d_model, h
        self.d_k = self.d_v = d_model // h
        for mod in ("vis", "ir"):
            setattr(self, f"que_proj_{mod}", nn.Linear(d_model, h * self.d_k))
            setattr(self, f"key_proj_{mod}", nn.Linear(d_model, h * self.d_k))
            setattr(self, f"val_proj_{mod}", nn.Linear(d_model, h * self.d_v))
        # registration order of the reference: 6 projections, then the two output projections
        self.out_proj_vis = nn.Linear(h * self.d_v, d_model)
        self.out_proj_ir = nn.Linear(h * self.d_v, d_model)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        self.LN1 = nn.LayerNorm(d_model)
        self.LN2 = nn.LayerNorm(d_model)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.001)
                nn.init.constant_(m.bias, 0)


def _mlp(d_model, block_exp, resid_pdrop):
    return nn.Sequential(nn.Linear(d_model, block_exp * d_model), nn.GELU(),
                         nn.Linear(block_exp * d_model, d_model), nn.Dropout(resid_pdrop))


class CrossTransformerBlock(HipModule):
    """One parameter-shared iterative cross-attention block (reference models/common.py:690-759).  `ln_input`,
    `ln_output`, `mlp` and the block-level `LN1` exist only so that reference checkpoints load strictly; the
    reference never uses them in forward."""

    def __init__(self, d_model, d_k, d_v, h, block_exp, attn_pdrop, resid_pdrop, loops_num=1):
        super().__init__()
        self.loops = loops_num
        self.ln_input = nn.LayerNorm(d_model)
        self.ln_output = nn.LayerNorm(d_model)
        self.crossatt = CrossAttention(d_model, d_k, d_v, h, attn_pdrop, resid_pdrop)
        self.mlp_vis = _mlp(d_model, block_exp, resid_pdrop)
        self.mlp_ir = _mlp(d_model, block_exp, resid_pdrop)
        self.mlp = _mlp(d_model, block_exp, resid_pdrop)
        self.LN1 = nn.LayerNorm(d_model)
        self.LN2 = nn.LayerNorm(d_model)
        for i in range(1, 9):
            setattr(self, f"coefficient{i}", LearnableCoefficient())

    # packed parameters: per-modality weight stacks [2][Np][Kp]
    def _packed(self, plan):
        def make():
            ca = self.crossatt
            dt, f = plan.dtype, (lambda t: t.detach().float())

            def stack(ws, bs):
                packs = [ops.pack_matrix(w, dt) for w in ws]
                wp = torch.stack([p[0] for p in packs]).contiguous()
                bp = torch.stack([ops.pack_bias(b, b.numel()) for b in bs]).contiguous()
                return wp, packs[0][1], bp
            qkv = stack([torch.cat([f(getattr(ca, f"{n}_proj_{m}").weight) for n in ("que", "key", "val")])
                         for m in ("vis", "ir")],
                        [torch.cat([f(getattr(ca, f"{n}_proj_{m}").bias) for n in ("que", "key", "val")])
                         for m in ("vis", "ir")])
            outp = stack([f(ca.out_proj_vis.weight), f(ca.out_proj_ir.weight)],
                         [f(ca.out_proj_vis.bias), f(ca.out_proj_ir.bias)])
            fc1 = stack([f(self.mlp_vis[0].weight), f(self.mlp_ir[0].weight)],
                        [f(self.mlp_vis[0].bias), f(self.mlp_ir[0].bias)])
            fc2 = stack([f(self.mlp_vis[2].weight), f(self.mlp_ir[2].weight)],
                        [f(self.mlp_vis[2].bias), f(self.mlp_ir[2].bias)])
            ln = {k: f(v).contiguous() for k, v in (("a1w", ca.LN1.weight), ("a1b", ca.LN1.bias),
                                                    ("a2w", ca.LN2.weight), ("a2b", ca.LN2.bias),
                                                    ("mw", self.LN2.weight), ("mb", self.LN2.bias))}
            co = [float(getattr(self, f"coefficient{i}").bias.detach().float().cpu()) for i in range(1, 9)]
            return dict(qkv=qkv, out=outp, fc1=fc1, fc2=fc2, ln=ln, co=co,
                        eps=(ca.LN1.eps, ca.LN2.eps, self.LN2.eps))
        return self._cached(("blk", plan.dtype, plan.device), make)

    @staticmethod
    def _gemm(plan, x, packed, y, cin, cout, act, res=None, alpha_acc=1.0, alpha_res=1.0, name="linear"):
        wp, kp, bp = packed
        rows = x.shape[1]
        gs = dict(x=x.stride(0), w=wp.stride(0), bias=bp.stride(0), y=y.stride(0),
                  res=res.stride(0) if res is not None else 0)

        def rows_act(t, c):       # group 0 of a (2, rows, C) token tensor (row stride t.stride(1)) as a (rows, 1, 1, c) act
            ld = t.stride(1)
            return t[0].as_strided((rows, 1, 1, c), (ld, ld, ld, 1))
        plan.add(ops.conv2d(rows_act(x, cin), wp, kp, bp, rows_act(y, cout), 1, 1, 1, 1, 0, 0,
                            cin, cout, act, res=rows_act(res, cout) if res is not None else None,
                            alpha_acc=alpha_acc, alpha_res=alpha_res, groups=2, group_strides=gs, name=name))

    # 16-bit types: one iteration = icaf_dmff_ln_qkv + icaf_dmff_attn_mlp (2 launches) instead of 7 (ICAF_DMFF_FUSE=0: A/B switch)
    fuse_block = OPT.dmff_fuse
    # The two-launch kernels (dmff_fused.hip: LN + QKV, then attention + out-projection + LN + MLP in ONE workgroup) are built for C <= 512 and
    # USED up to this width.  Round 2 measured them ahead at C <= 128 only (111 vs 145 us for the seven per-layer launches at P3; at C = 256 /
    # 512 one workgroup per CU cannot hide its own latencies: 166 vs 141, 430 vs 153 us); round 3 moved C = 256 / 512 to the three-launch form
    # (dmff_wide.hip); round 4 built that form for C = 128 too (four wavefronts, weights streamed per wave into registers: no barrier in a
    # GEMM pass) and the stand-alone attention kernel got the rewritten inner loop: P3 block 88 us (two launches) vs 78 us (three), whole
    # bench 15,003 / 15,061 vs 15,181 pairs/s on one box — so the default is 64 and every yolov5s level runs three launches.
    # ICAF_DMFF_FUSE_MAX_C=128 restores the two-launch form at P3 (A/B, tests).
    fuse_max_c = OPT.dmff_fuse_max_c
    # fp32 plans keep the per-layer launches (the goldens then cover them); True runs the fp32 INSTANTIATION of the fused kernels where
    # it exists: the same templates the 16-bit path runs, held to the reference's fp32 goldens (tests/test_gpu_dmff_fused.py) — the two-launch
    # form at C <= fuse_max_c <= 128, and (round 5) the THREE-launch form every yolov5s level runs by default, at C = 128 (dmff_wide.hip)
    fuse_fp32 = OPT.dmff_fuse_fp32

    def fusable(self, plan, C, N):
        hid, h = self.mlp_vis[0].out_features, self.crossatt.h
        types = (torch.bfloat16, torch.float16, torch.float32) if (self.fuse_fp32 and C <= 128) else (torch.bfloat16, torch.float16)
        return (self.fuse_block and C <= self.fuse_max_c and plan.dtype in types and C % 64 == 0 and (C // h) % 8 == 0
                and hid % 128 == 0 and self.mlp_vis[2].in_features == hid
                and (plan.device.type != "cuda" or ops.dmff_fused_lds_bytes(C, N, h, plan.dtype) is not None)
                and (plan.device.type == "cuda" or C <= 512))

    # wide levels, 16-bit types: three launches per iteration (ICAF_DMFF_WIDE=0: the seven per-layer launches, A/B switch)
    fuse_wide = OPT.dmff_wide
    res32 = OPT.dmff_res32        # loops > 1: the token stream between iterations in fp32 (A/B switch)
    wide_max_c = OPT.dmff_wide_max_c

    def wide_fusable(self, plan, C):
        hid, h = self.mlp_vis[0].out_features, self.crossatt.h
        if plan.dtype == torch.float32 and not self.fuse_fp32:         # fp32 plans keep the per-layer launches unless the parity instantiation is asked for
            return False
        return (self.fuse_wide and self.fuse_block and self.fuse_max_c < C <= self.wide_max_c and ops.dmff_wide_ok(C, hid, plan.dtype)
                and (C // h) % 8 == 0 and self.mlp_vis[2].in_features == hid)

    def emit_tokens(self, plan, tok, B, N, final_out=None):
        """tok: (2, B*N, C) [0]=RGB [1]=IR -> same shape after `loops` shared-weight iterations.  final_out: optional
        (2, B*N, C) view (any row / group strides) that the last iteration writes instead of a fresh buffer."""
        p = self._packed(plan)
        C = tok.shape[2]
        rows = tok.shape[1]
        co, ln = p["co"], p["ln"]
        hid = self.mlp_vis[0].out_features
        nloops = int(self.loops)
        if self.fusable(plan, C, N):
            # fused block (dmff_fused.hip): LayerNorm + QKV, then attention + out-projection + LayerNorm + MLP; attention output,
            # x_att, the normalised tile and the hidden activations never reach HBM
            coef = dict(co=co, hidden=hid)
            qkv = plan.tokens(2, rows, 3 * C)
            for it in range(nloops):
                plan.add(ops.dmff_ln_qkv(tok, qkv, p, ln, coef, p["eps"], B, N, self.crossatt.h))
                nxt = final_out if (final_out is not None and it == nloops - 1) else plan.tokens(2, rows, C)
                plan.add(ops.dmff_attn_mlp(tok, qkv, nxt, p, ln, coef, p["eps"], B, N, self.crossatt.h))
                tok = nxt
            if nloops > 1 and plan.dtype != torch.float32:
                plan.notes.setdefault("dmff_fp32_token_stream", {})[f"C={C}"] = False
            return tok
        if self.wide_fusable(plan, C):
            # wide levels (C = 256 / 512): LayerNorm + QKV, attention, out-projection + LayerNorm + MLP — three launches; x_att, the
            # normalised tile and the hidden activations never reach HBM (dmff_wide.hip)
            coef = dict(co=co, hidden=hid)
            qkv = plan.tokens(2, rows, 3 * C)
            att = plan.tokens(2, rows, C)
            ks = ops.dmff_wide_ksplit(N, C, hid)                   # few tokens per image, weights beyond an XCD's L2: hidden columns split over ks workgroups
            part = plan.empty((ks, 2, rows, C), torch.float32) if ks > 1 else None
            # several iterations: the residual chain x -> x_att -> x' is carried in FP32 from one iteration to the next (two ping-pong buffers); the
            # 16-bit tokens are still written (LayerNorm + QKV read them).  Rounding the stream to 16 bits twice per iteration had used up the parity
            # margin of the 3-iteration configuration (0.91 x the reference's own bf16 error)
            r32 = self.res32 and nloops > 1 and plan.dtype != torch.float32 and ks in (1, 2) and not (C == 512 and ks == 1)      # (not built: dmff_wide.hip)
            t32 = [plan.empty((2, rows, C), torch.float32) for _ in range(2)] if r32 else None
            if nloops > 1:
                plan.notes.setdefault("dmff_fp32_token_stream", {})[f"C={C}"] = bool(r32)
            # (the LAST iteration writes y32 too although nothing reads it — 2 * rows * C * 4 bytes, 3 % of the launch's traffic at P3: the kernels' R32
            #  instantiation writes both token streams or neither, and a third instantiation for one launch in `loops` was not worth its compile time)
            for it in range(nloops):
                plan.add(ops.dmff_wide_ln_qkv(tok, qkv, p, ln, coef, p["eps"], B, N, self.crossatt.h))
                plan.add(ops.cross_attention(qkv, att, B, N, self.crossatt.h))
                nxt = final_out if (final_out is not None and it == nloops - 1) else plan.tokens(2, rows, C)
                ls = ops.dmff_wide_proj_mlp(tok, att, nxt, p, ln, coef, p["eps"], B, N, self.crossatt.h, partial=part, ksplit=ks,
                                            x32=(t32[(it - 1) & 1] if (r32 and it > 0) else None), y32=(t32[it & 1] if r32 else None))
                for l in (ls if isinstance(ls, list) else [ls]):
                    plan.add(l)
                tok = nxt
            return tok
        if nloops > 1 and plan.dtype != torch.float32:
            plan.notes.setdefault("dmff_fp32_token_stream", {})[f"C={C}"] = False
        for it in range(nloops):
            n1 = plan.tokens(2, rows, C)
            plan.add(ops.layernorm(tok, n1, ln["a1w"], ln["a1b"], ln["a2w"], ln["a2b"], p["eps"][0], name="ln_attn"))
            qkv = plan.tokens(2, rows, 3 * C)
            self._gemm(plan, n1, p["qkv"], qkv, C, 3 * C, ops.ACT_NONE, name="qkv_proj")
            att = plan.tokens(2, rows, C)
            plan.add(ops.cross_attention(qkv, att, B, N, self.crossatt.h))
            xatt = plan.tokens(2, rows, C)
            self._gemm(plan, att, p["out"], xatt, C, C, ops.ACT_NONE, res=tok, alpha_res=(co[0], co[2]),
                       alpha_acc=(co[1], co[3]), name="out_proj")
            n2 = plan.tokens(2, rows, C)
            plan.add(ops.layernorm(xatt, n2, ln["mw"], ln["mb"], ln["mw"], ln["mb"], p["eps"][2], name="ln_mlp"))
            h = plan.tokens(2, rows, hid)
            self._gemm(plan, n2, p["fc1"], h, C, hid, ops.ACT_GELU, name="mlp_fc1")
            nxt = final_out if (final_out is not None and it == nloops - 1) else plan.tokens(2, rows, C)
            self._gemm(plan, h, p["fc2"], nxt, hid, C, ops.ACT_NONE, res=xatt, alpha_res=(co[4], co[6]),
                       alpha_acc=(co[5], co[7]), name="mlp_fc2")
            tok = nxt
        return tok


class TransformerFusionBlock(HipModule):
    """DMFF: Dual-Modality Feature Fusion (reference models/common.py:762-865).

    The reference's positional signature is kept; the extra keyword `loops_num` (yaml: a trailing `{loops_num: n}` mapping,
    see models/yolo.py) exposes the iterative parameter-shared loop the reference wires but never surfaces
    (models/common.py:691,744)."""

    # class-level (not set in __init__): instances un-pickled from reference checkpoints never ran this __init__
    fuse_tail = True         # run interpolate + residual + cat + conv1x1_out as one GEMM when the layout allows

    def __init__(self, d_model, vert_anchors=16, horz_anchors=16, h=8, block_exp=4, n_layer=1, embd_pdrop=0.1,
                 attn_pdrop=0.1, resid_pdrop=0.1, loops_num=1):
        super().__init__()
        self.n_embd, self.vert_anchors, self.horz_anchors = d_model, vert_anchors, horz_anchors
        self.pos_emb_vis = nn.Parameter(torch.zeros(1, vert_anchors * horz_anchors, d_model))
        self.pos_emb_ir = nn.Parameter(torch.zeros(1, vert_anchors * horz_anchors, d_model))
        self.avgpool = AdaptivePool2d(vert_anchors, horz_anchors, "avg")
        self.maxpool = AdaptivePool2d(vert_anchors, horz_anchors, "max")
        self.vis_coefficient = LearnableWeights()
        self.ir_coefficient = LearnableWeights()
        self.crosstransformer = nn.Sequential(*[
            CrossTransformerBlock(d_model, d_model, d_model, h, block_exp, attn_pdrop, resid_pdrop, loops_num)
            for _ in range(n_layer)])
        self.concat = Concat(dimension=1)
        self.conv1x1_out = Conv(c1=d_model * 2, c2=d_model, k=1, s=1, p=0, g=1, act=True)

    def emit(self, plan, xs, out=None):
        rgb, ir = xs
        B, H, W, C = rgb.shape
        assert ir.shape == rgb.shape and C == self.n_embd
        th, tw, kh, kw, sh, sw = self.avgpool.window(H, W)
        N = th * tw
        if N != self.pos_emb_vis.shape[1]:
            raise ValueError(f"DMFF: {th}x{tw} tokens do not match the {self.pos_emb_vis.shape[1]} positional "
                             "embeddings (input feature map smaller than the anchor grid)")

        def make():
            f = lambda t: t.detach().float().reshape(-1).contiguous()          # noqa: E731
            wv, wi = self.vis_coefficient, self.ir_coefficient
            return (f(self.pos_emb_vis), f(self.pos_emb_ir),
                    (float(wv.w1.detach().float().cpu()), float(wv.w2.detach().float().cpu())),
                    (float(wi.w1.detach().float().cpu()), float(wi.w2.detach().float().cpu())))
        pos_v, pos_i, w_v, w_i = self._cached(("dmff", plan.device), make)
        tok = plan.tokens(2, B * N, C)
        plan.add(ops.dmff_pool_tokens(rgb, ir, pos_v, pos_i, tok, th, tw, kh, kw, sh, sw, w_v, w_i))
        feat2c = concat_view([rgb, ir]) if self.fuse_tail else None
        if (2 * C * rgb.element_size()) % 128:        # the pre-term GEMM instantiations walk K in whole 128-byte slices
            feat2c = None
        blocks = list(self.crosstransformer)
        if feat2c is None or not blocks or int(blocks[-1].loops) < 1:
            # general path: materialise cat(rgb + up(tok_rgb), ir + up(tok_ir)), then the 1x1 fuse conv
            for blk in blocks:
                tok = blk.emit_tokens(plan, tok, B, N)
            merged = plan.act(B, H, W, 2 * C)
            plan.add(ops.dmff_upsample_merge(tok, rgb, ir, merged, th, tw))
            return self.conv1x1_out.emit(plan, merged, out=out)
        # Fused tail (both streams' features are adjacent channel slices of one buffer, as Model.build_plan places them):
        # the 1x1 convolution commutes with the bilinear resize and with the residual add (all linear), so
        #   conv(cat(rgb + up(t_rgb), ir + up(t_ir))) = conv(cat(rgb, ir)) + up(conv(cat(t_rgb, t_ir)))
        # The token-resolution product P is a tiny GEMM (fp32 out); the full-resolution GEMM reads the untouched
        # features and adds bilinear(P) before bias + SiLU in its epilogue.  The (B, H, W, 2C) merged tensor of
        # models/common.py:827-840 is never written or read.
        tokcat = plan.empty((B * N, 2 * C), plan.dtype)                      # [t_rgb | t_ir] per token
        final = tokcat.as_strided((2, B * N, C), (C, 2 * C, 1))
        for j, blk in enumerate(blocks):
            tok = blk.emit_tokens(plan, tok, B, N, final_out=final if j == len(blocks) - 1 else None)
        conv = self.conv1x1_out

        def make():
            w, _ = conv.folded()                                              # BN scale folded in; bias stays in the main GEMM
            wp, kp = ops.pack_conv_weight(w, plan.dtype)
            return wp, kp
        wp, kp = self._cached(("tailw", plan.dtype, plan.device), make)
        P = plan.empty((B, th, tw, C), torch.float32)
        plan.add(ops.conv2d(tokcat.view(B * N, 1, 1, 2 * C), wp, kp, None, P.view(B * N, 1, 1, C), 1, 1, 1, 1, 0, 0,
                            2 * C, C, ops.ACT_NONE, name="dmff_tail_tokens"))
        return conv.emit(plan, feat2c, out=out, pre_term=P)


# ----------------------------------------------------------------------------------------------------------
# Detect
# ----------------------------------------------------------------------------------------------------------
class Detect(HipModule):
    """Detection head (reference models/yolo_test.py:26-71).  Per level: 1x1 conv (fp32 output) + one decode
    kernel that writes z / logits / the permuted raw map directly in their final layouts."""
    stride = None
    export = False
    # 16-bit plans: a level's 1x1 conv and its decode run as ONE launch (icaf_detect_conv; ICAF_DETECT_FUSE=0: A/B switch).  A class
    # default: un-pickled reference checkpoints never run this constructor.
    fuse_decode = OPT.detect_fuse

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.nl, self.na = len(anchors), len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in ch)

    def _apply(self, fn, *a, **k):
        """Anchors stay fp32 whatever the model is cast to: pixel sizes such as 373 are not representable in bf16
        and the decode runs in fp32 anyway."""
        keep = {n: getattr(self, n).detach().float().clone() for n in ("anchors", "anchor_grid")}
        r = super()._apply(fn, *a, **k)
        for n, v in keep.items():
            cur = getattr(self, n)
            if cur.dtype != torch.float32:
                setattr(self, n, v.to(cur.device))
        return r

    def emit(self, plan, xs):
        B = xs[0].shape[0]
        rows = sum(self.na * t.shape[1] * t.shape[2] for t in xs)
        z = plan.empty((B, rows, self.no), torch.float32)
        logits = plan.empty((B, rows, self.nc), torch.float32)
        raws, off = [], 0
        ag = self.anchor_grid.detach().float().cpu().view(self.nl, self.na * 2)
        strides = [float(s) for s in self.stride]
        for l, x in enumerate(xs):
            _, ny, nx, c = x.shape
            conv = self.m[l]
            nout = self.na * self.no

            def make(conv=conv):
                wp, kp = ops.pack_conv_weight(conv.weight.detach().float(), plan.dtype)
                return wp, kp, ops.pack_bias(conv.bias.detach().float(), conv.out_channels)
            wp, kp, bp = self._cached(("det", l, plan.dtype, plan.device), make)
            raw = plan.empty((B, self.na, ny, nx, self.no), torch.float32)
            if self.fuse_decode and ops.detect_conv_ok(x, self.na, self.no, c):
                # 16-bit maps: conv + decode in ONE launch (icaf_detect_conv) - the fp32 conv map never reaches HBM
                plan.add(ops.detect_conv(x, wp, kp, bp, z, logits, raw, self.na, self.no, off, strides[l], ag[l].tolist(), c))
                raws.append(raw)
                off += self.na * ny * nx
                continue
            # pixel stride padded to a 16-byte multiple (18 -> 20 floats): the conv epilogue then leaves 16-byte stores instead of scalar ones
            p = plan.act(B, ny, nx, (nout + 3) // 4 * 4, dtype=torch.float32)[..., :nout]
            plan.add(ops.conv2d(x, wp, kp, bp, p, 1, 1, 1, 1, 0, 0, c, nout, ops.ACT_NONE, name="detect_conv"))
            plan.add(ops.detect_decode(p, z, logits, raw, self.na, self.no, off, strides[l], ag[l].tolist()))
            raws.append(raw)
            off += self.na * ny * nx
        return z, logits, raws

    def forward(self, xs):
        if self.training:
            raise NotImplementedError("Detect: training branch is outside the inference hot path")
        dt = _module_dtype(self)
        plan = Plan(xs[0].device, dt)
        acts = [to_act(t, dt) for t in xs]
        z, logits, raws = self.emit(plan, acts)
        plan.run()
        return z, logits, raws


__all__ = ["autopad", "Conv", "Bottleneck", "C3", "SPPF", "Concat", "Add", "NiNfusion", "LearnableCoefficient", "LearnableWeights",
           "AdaptivePool2d", "CrossAttention", "CrossTransformerBlock", "TransformerFusionBlock", "Detect",
           "emit_upsample", "HipModule", "math"]
