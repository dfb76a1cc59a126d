"""Thin torch-tensor front end over the C ABI (include/icaf.h).

Activation tensors ("acts") are torch views of shape (B, H, W, C) whose last dim is contiguous and whose pixel
stride ld = t.stride(2) may exceed C (channel slice of a wider NHWC buffer).  Every function here either launches
one kernel on a stream or returns a `Launch` record that does so later (used by the static execution plan).
PyTorch only provides device memory and streams here — none of its operators run on the hot path.
"""
import ctypes as C
import weakref

import torch

from . import _lib
from .options import OPT
from ._lib import BneckArgs, ConvArgs, DmffArgs, Stem2Args, check, lib, F32, BF16, F16, ACT_NONE, ACT_SILU, ACT_GELU  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
VEC = {torch.float32: 4, torch.bfloat16: 8, torch.float16: 8}     # elements per 16-byte vector
N_ALIGN = 128     # packed-weight rows are padded to this (largest BN tile)
K_ALIGN_BYTES = 128


def dtype_code(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise _lib.IcafError(f"unsupported dtype {dt}; supported: float32, bfloat16, float16") from None


def k_align(dt):
    return K_ALIGN_BYTES // torch.empty((), dtype=dt).element_size()


def current_stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _act_geom(t):
    """(B, H, W, C, ld) of an act; a 5-D (2, B, H, W, C) *pair* (the two streams of the backbone at a constant
    element stride, t.stride(0)) reports the geometry of one member."""
    if t.dim() == 5:
        assert t.shape[0] == 2, f"pair acts hold exactly two streams, got {t.shape}"
        t = t[0]
    assert t.dim() == 4 and t.stride(3) == 1, f"act must be NHWC-contiguous in C, got {t.shape} {t.stride()}"
    B, H, W, Cc = t.shape
    ld = t.stride(2)
    assert t.stride(1) == W * ld and (B == 1 or t.stride(0) == H * W * ld), f"bad act strides {t.stride()}"
    return B, H, W, Cc, ld


def flat_pair(t):
    """View a pair act (2, B, H, W, C) as one act of batch 2B (weight-less kernels treat the streams as more batch)."""
    if t.dim() == 4:
        return t
    G, B, H, W, Cc = t.shape
    assert t.stride(0) == B * t.stride(1), "pair members must be adjacent in memory"
    return t.as_strided((G * B, H, W, Cc), t.stride()[1:])


class Launch:
    """One recorded kernel launch: fn(*args, stream)."""
    __slots__ = ("fn", "args", "keep", "name", "flops", "bytes", "branch")

    def __init__(self, fn, args, keep=(), name="", flops=0, nbytes=0):
        self.fn, self.args, self.keep, self.name, self.flops, self.bytes = fn, args, keep, name, flops, nbytes
        self.branch = 0          # 0 = main chain; > 0 = a side branch of the captured graph (engine.Plan.branches)

    def __call__(self, stream_ptr):
        st = self.fn(*self.args, stream_ptr)
        if st != 0:
            check(st, self.name)


# ------------------------------------------------------------------------------------------------------------
# weight packing (one-time, at plan build; not on the hot path)
# ------------------------------------------------------------------------------------------------------------
def pack_matrix(w2d, dt):
    """[Cout][K] fp32 -> zero-padded [Np][Kp] in dtype dt (K-major rows)."""
    n, k = w2d.shape
    ka = k_align(dt)
    np_, kp = -(-n // N_ALIGN) * N_ALIGN, -(-k // ka) * ka
    out = torch.zeros((np_, kp), dtype=dt, device=w2d.device)
    out[:n, :k] = w2d.to(dt)
    return out, kp


def pack_bias(b, n):
    np_ = -(-n // N_ALIGN) * N_ALIGN
    out = torch.zeros((np_,), dtype=torch.float32, device=b.device)
    out[:n] = b.float()
    return out


def pack_conv_weight(w4d, dt, cin_pad=None):
    """[Cout][Cin][kh][kw] -> [Np][Kp] with k = (kh, kw, cin) and cin padded to cin_pad."""
    co, ci, kh, kw = w4d.shape
    w = w4d.permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad != ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))
    return pack_matrix(w.reshape(co, -1).contiguous(), dt)


_FRAG_CACHE = {}              # id(packed tensor) -> (weakref to it, fragment-major copy); entries die with the packed tensor
# (execution switches — which kernels the tuner may offer a layer, retune requests — live in options.PlanOptions; read as OPT.<field>)
# (Round 5's persistent long-K GEMM — launch configuration 67, igemm_pers.hip — and the persistent halo-patch 3x3 — 90 + shape, cwpers.hip — were removed
#  in round 6: the first won isolated and lost every benchmarked workload with two forwards in flight (yolov5l shard 3,813 -> 3,778 pairs/s), the second
#  lost everywhere; no committed tune cache named either.  `git log -- icafusion_amd/csrc/igemm_pers.hip` has them; docs/HISTORY.md section 16 the numbers.)


def frag_weights(w_packed):
    """Second copy of packed 16-bit weights ([Np][Kp] or stacked [G][Np][Kp]) in FRAGMENT-MAJOR order for the kernels that feed
    the weight operand from registers (igemm_wreg.hip; icaf.h: icaf_conv_args.wf): [G][Np / 32][Kp / 16][64][8], lane (hi * 32 + r) of
    block (nb, ks) = w[nb * 32 + r][ks * 16 + hi * 8 : + 8].  Built once per packed tensor (plan-build time), cached by storage."""
    key = id(w_packed)
    hit = _FRAG_CACHE.get(key)
    if hit is not None and hit[0]() is w_packed:
        return hit[1]
    w = w_packed if w_packed.dim() == 3 else w_packed[None]
    G, np_, kp = w.shape
    ve = 16 // w.element_size()                     # elements per 16-byte fragment: 8 (16-bit types), 4 (fp32: the parity instantiation of dmff_wide.hip)
    assert np_ % 32 == 0 and kp % (2 * ve) == 0 and w.element_size() in (2, 4)
    f = w.reshape(G, np_ // 32, 32, kp // (2 * ve), 2, ve).permute(0, 1, 3, 4, 2, 5).contiguous()      # g, nb, ks, hi, r, e
    f = f.reshape(G, np_ // 32, kp // (2 * ve), 64, ve)
    if w_packed.dim() == 2:
        f = f[0]
    # The copy lives exactly as long as the packed tensor it mirrors: HipModule.invalidate() / .to() / load_state_dict drop the module's
    # packed tensors, and the finaliser then drops the copy (a strong reference here pinned every old pack in device memory forever).
    _FRAG_CACHE[key] = (weakref.ref(w_packed), f)
    weakref.finalize(w_packed, _FRAG_CACHE.pop, key, None)
    return f


def s2d_conv_weight(w4d):
    """6x6/s2/p2 kernel over C channels -> equivalent 3x3/s1/p1 kernel over the 4C space-to-depth channels
    ordered (dy, dx, c):  W3[co][(dy*2+dx)*C + c][ty][tx] = W[co][c][2ty+dy][2tx+dx]."""
    co, ci, kh, kw = w4d.shape
    assert kh == 6 and kw == 6
    w = w4d.reshape(co, ci, 3, 2, 3, 2)                  # co, c, ty, dy, tx, dx
    return w.permute(0, 3, 5, 1, 2, 4).reshape(co, 4 * ci, 3, 3).contiguous()


# ------------------------------------------------------------------------------------------------------------
# launches
# ------------------------------------------------------------------------------------------------------------
def conv2d(x, w_packed, kp, bias, y, kh, kw, sh, sw, ph, pw, cin, cout, act, res=None, alpha_acc=1.0,
           alpha_res=1.0, groups=1, group_strides=None, tile=0, name="conv2d", pre=None, chain=None, pre_nearest=False):
    """Record an implicit-GEMM conv / linear.  x, y, res are acts; for groups=2 they are the group-0 views and
    group_strides = dict(x=, w=, bias=, y=, res=) gives element strides to group 1."""
    B, H, W, cx, ldx = _act_geom(x)
    By, Ho, Wo, cy, ldy = _act_geom(y)
    assert cx >= cin and cy >= cout and By == B
    assert Ho == (H + 2 * ph - kh) // sh + 1 and Wo == (W + 2 * pw - kw) // sw + 1, "conv geometry mismatch"
    if x.dim() == 5:                  # pair act: both backbone streams in one launch, per-stream weights
        assert group_strides is None and y.dim() == 5 and w_packed.dim() == 3 and (res is None or res.dim() == 5)
        groups = 2
        group_strides = dict(x=x.stride(0), w=w_packed.stride(0), bias=bias.stride(0) if bias is not None else 0,
                             y=y.stride(0), res=res.stride(0) if res is not None else 0)
    a = ConvArgs()
    a.x, a.w, a.y = x.data_ptr(), w_packed.data_ptr(), y.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    gs = group_strides or {}
    a.x_gs, a.w_gs, a.bias_gs = gs.get("x", 0), gs.get("w", 0), gs.get("bias", 0)
    a.y_gs, a.res_gs = gs.get("y", 0), gs.get("res", 0)
    a.groups = groups
    a.B, a.H, a.W, a.Cin, a.ldx = B, H, W, cin, ldx
    a.Ho, a.Wo, a.Cout, a.ldy = Ho, Wo, cout, ldy
    a.kh, a.kw, a.sh, a.sw, a.ph, a.pw = kh, kw, sh, sw, ph, pw
    a.ldr = _act_geom(res)[4] if res is not None else 0
    a.Kp, a.act = kp, act
    a.dtype, a.out_dtype = dtype_code(x.dtype), dtype_code(y.dtype)
    assert w_packed.dtype == x.dtype and (res is None or res.dtype == x.dtype)
    aa = alpha_acc if isinstance(alpha_acc, (tuple, list)) else (alpha_acc, alpha_acc)
    ar = alpha_res if isinstance(alpha_res, (tuple, list)) else (alpha_res, alpha_res)
    a.alpha_acc[0], a.alpha_acc[1] = float(aa[0]), float(aa[1])
    a.alpha_res[0], a.alpha_res[1] = float(ar[0]), float(ar[1])
    a.tile = tile
    wf = None
    cw_layer = bool(OPT.cwide and act == ACT_SILU and cwide_shapes(kh, kw, sh, sw, ph, pw, cin, cout))                               # cwide.hip
    if (x.dtype != torch.float32 and y.dtype == x.dtype and (cin * 2) % 128 == 0 and kp % 64 == 0 and pre is None
            and ((OPT.wreg_gemm and chain is None and cout > 64) or cw_layer)):
        wf = frag_weights(w_packed)                   # igemm_wreg.hip: weight operand from registers (tile ids 61 / 62)
        a.wf, a.wf_gs = wf.data_ptr(), (wf.stride(0) if w_packed.dim() == 3 else 0)
    if pre is not None:               # fp32 coarse map (B, h, w, >= cout) added, bilinearly resized, before the activation
        Bp, hp, wp_, cp, ldp = _act_geom(pre)
        assert pre.dtype == torch.float32 and Bp == B and cp >= cout and groups == 1
        a.pre, a.pre_h, a.pre_w, a.ldpre = pre.data_ptr(), hp, wp_, ldp
        a.pre_mode = int(bool(pre_nearest))     # nearest: the map is the low-resolution half of a 1x1 conv over cat(up(a), b)
    if chain is not None:             # chained 1x1 + SiLU on the output tile (icaf.h): dict(w=, kp=, bias=, y=, cout=[, keep=][, x2=])
        y2 = chain["y"]
        B2, H2, W2, c2y, ldy2 = _act_geom(y2)
        assert (B2, H2, W2) == (B, Ho, Wo) and c2y >= chain["cout"] and y2.dtype == x.dtype and (y2.dim() == 5) == (x.dim() == 5)
        a.w2, a.y2 = chain["w"].data_ptr(), y2.data_ptr()
        a.bias2 = chain["bias"].data_ptr() if chain.get("bias") is not None else None
        a.Kp2, a.Cout2, a.ldy2 = chain["kp"], chain["cout"], ldy2
        a.chain_keep = int(bool(chain.get("keep")))
        if x.dim() == 5:
            a.w2_gs, a.y2_gs = chain["w"].stride(0), y2.stride(0)
            a.bias2_gs = chain["bias"].stride(0) if chain.get("bias") is not None else 0
        x2 = chain.get("x2")
        if x2 is not None:            # C3 tail: the chained layer reads K = [this layer's tile | x2] (icaf.h: icaf_conv_args.x2)
            Bx, Hx, Wx, cx2, ldx2 = _act_geom(x2)
            assert (Bx, Hx, Wx) == (B, Ho, Wo) and cx2 >= cout and x2.dtype == x.dtype and (x2.dim() == 5) == (x.dim() == 5)
            a.x2, a.ldx2, a.x2_gs = x2.data_ptr(), ldx2, (x2.stride(0) if x.dim() == 5 else 0)
    m = B * Ho * Wo
    flops = 2.0 * m * cout * kh * kw * cin * groups
    es, eo = x.element_size(), y.element_size()
    nbytes = groups * (B * H * W * cin * es + cout * kh * kw * cin * es + m * cout * eo
                       + (m * cout * es if res is not None else 0))
    if chain is not None:
        k2 = cout * (2 if chain.get("x2") is not None else 1)
        flops += 2.0 * m * k2 * chain["cout"] * groups
        nbytes += groups * (m * chain["cout"] - (0 if chain.get("keep") else m * cout)) * eo      # y2 is written instead of / besides y
        if chain.get("x2") is not None:
            nbytes += groups * (m * cout + k2 * chain["cout"]) * es       # the x2 half of the chained layer's input, its weights
    return Launch(lib().icaf_conv2d, (C.byref(a),), keep=(a, x, w_packed, bias, y, res, pre, chain, wf), name=name, flops=flops,
                  nbytes=nbytes)


def bottleneck(x, w1_packed, kp1, bias1, w2_packed, kp2, bias2, y, c, add, shape, name="bottleneck", cv3=None):
    """Whole Bottleneck (1x1 -> 3x3 [+ x]) in one launch (icaf_bottleneck).  x, y: acts or pair acts of c channels in
    DIFFERENT buffers; w1 / w2: packed 1x1 / 3x3 weights (stacked per stream for pair acts).
    cv3 = dict(w=, kp=, bias=, y=, cout=, x2=): the C3's cv3 rides on the block — x2 is the cv2 half of its input, w its
    packed weights with K columns ordered [cv2 | m]; only cv3's output (y of the dict) is written and `y` may be None."""
    if cv3 is not None and y is None:
        y = cv3["y"][..., :c]                     # (ignored by the kernel; satisfies the argument checks)
    inner = conv2d(x, w2_packed, kp2, bias2, y, 3, 3, 1, 1, 1, 1, c, c, ACT_SILU, res=x if add else None, name=name,
                   chain=None if cv3 is None else dict(w=cv3["w"], kp=cv3["kp"], bias=cv3["bias"], y=cv3["y"], cout=cv3["cout"]))
    b = BneckArgs()
    C.memmove(C.byref(b.conv), C.byref(inner.keep[0]), C.sizeof(ConvArgs))
    b.w1, b.bias1 = w1_packed.data_ptr(), bias1.data_ptr()
    paired = x.dim() == 5
    b.w1_gs, b.bias1_gs = (w1_packed.stride(0), bias1.stride(0)) if paired else (0, 0)
    b.Kp1, b.shape = kp1, shape
    B, H, W, _, _ = _act_geom(x)
    g = 2 if paired else 1
    flops = inner.flops + 2.0 * g * B * H * W * c * c
    es = x.element_size()
    nbytes = g * (B * H * W * c * es * (3 if add else 2) + (9 * c * c + c * c) * es)
    if cv3 is not None:
        x2 = cv3["x2"]
        B2, H2, W2, c2, ldx2 = _act_geom(x2)
        assert (B2, H2, W2) == (B, H, W) and c2 >= c and x2.dtype == x.dtype and (x2.dim() == 5) == paired
        b.x2, b.ldx2, b.x2_gs = x2.data_ptr(), ldx2, x2.stride(0) if paired else 0
        nbytes = g * (B * H * W * es * (2 * c + cv3["cout"]) + (10 * c * c + 2 * c * cv3["cout"]) * es)
        flops += 2.0 * g * B * H * W * c * cv3["cout"]          # (conv2d counted K = c for the chained layer; cv3 has K = 2c)
    return Launch(lib().icaf_bottleneck, (C.byref(b),), keep=(b, inner.keep, w1_packed, bias1, cv3), name=name + ("+cv3" if cv3 else ""),
                  flops=flops, nbytes=nbytes)


_TUNE_CACHE = {}
_RETUNED = set()
CTILE_SHAPES = {1: (32, 1), 2: (64, 1), 3: (64, 1), 4: (64, 2), 5: (128, 1)}     # shape id -> (BN, stride), ctile.hip
CONV_PIPELINES = (0, 1, 2)        # LDS-DMA 64 B x3, register-staged, LDS-DMA 128 B x2 (3 = 128 B x3: never won)


def cwide_shapes(kh, kw, sh, sw, ph, pw, cin, cout):
    """Tile ids (80 + shape) of cwide.hip that are built for this 3x3 layer: resident halo patch, weights streamed into registers."""
    if (kh, kw, ph, pw) != (3, 3, 1, 1) or sh != sw or cout % 128:
        return []
    # 80 + shape: one tile per workgroup
    if sh == 1:
        return [81, 82] if (cin == 128 and cout == 128) else []          # 8 x 16 / 8 x 8 output pixels per workgroup
    if sh == 2:
        if cin == 64:
            return [83, 85]                                              # stride 2, 64 -> 128 k: 8 x 16 / 8 x 8
        if cin == 128:
            return [84]                                                  # stride 2, 128 -> 128 k: 8 x 8
    return []


def _conv_signature(a):
    return (a.B * a.Ho * a.Wo, a.Cout, a.Cin, a.kh, a.kw, a.sh, a.sw, a.H, a.W, a.ldx, a.ldy, a.groups, a.dtype,
            a.out_dtype, a.act, bool(a.res), bool(a.pre) + a.pre_mode, a.Cout2 if a.w2 else 0, 2 if a.x2 else bool(a.chain_keep))


def conv_candidates(a):
    """Launch-configuration ids worth timing for one conv (ConvArgs `a`): igemm tiles x pipelines, the 8-wavefront tiles,
    the 3x3 halo-patch kernel; chained / pre-term launches only have the configurations that are built for them."""
    cands = []
    cs_ok = ((a.kh, a.kw, a.sh, a.sw, a.ph, a.pw) == (3, 3, 1, 1, 1, 1) and a.Cin == 64 and a.Cout <= 64 and a.Cout % 8 == 0 and a.dtype != F32
             and a.out_dtype == a.dtype and a.act == ACT_SILU and not a.pre and OPT.cstream)
    cw = (cwide_shapes(a.kh, a.kw, a.sh, a.sw, a.ph, a.pw, a.Cin, a.Cout)
          if (a.dtype != F32 and a.wf and a.out_dtype == a.dtype and a.act == ACT_SILU and not a.pre and OPT.cwide) else [])
    if a.w2 and a.x2:                  # C3 tail (the chained cv3 reads [tile | x2]): cwide.hip's 8 x 16 / 8 x 8 forms only
        # Below ~200 k pixels per stream (the 40 x 40 maps of yolov5s at batch 32 / 64: one round of 8 x 16 tiles for the chip) only the 8 x 8
        # form is offered: isolated timings prefer 8 x 16 there (2 workgroups of 252 registers and 70 KB per CU), but with a second forward in
        # flight that form starves the co-running kernels — same-box A/B of the whole bench: 15,858 with 8 x 16 against 16,082 without the tail and
        # 16,091 with 8 x 8 (3 workgroups of 168 registers and 35 KB).  The 80 x 80 / 160 x 160 maps of yolov5l have 4 - 16 x the tiles: tuned.
        cands = ([82] if a.B * a.Ho * a.Wo < OPT.tail_8x16_minpix else [81, 82]) if OPT.cwide else []
    elif a.w2:                         # chained 1x1: one N tile covering both layers, LDS-DMA pipelines 0 / 2
        t = 2 if max(a.Cout, a.Cout2) <= 64 else 1
        cands = [t, t + 20]
        if cw and a.Cout == 128 and a.Cout2 <= 128 and a.Cout2 % 32 == 0:
            cands += cw                                                      # resident halo patch, weights (and the chained 1x1's) streamed into registers (cwide.hip)
        if cs_ok and a.Cout == 64 and a.Cout2 <= 64:
            cands.append(71)                 # persistent 3x3 with the filter (and the chained 1x1) resident in LDS (cstream.hip)
    elif a.pre:                          # pre-activation term: built for tiles 128x128 / 128x64 on the LDS-DMA pipelines 0 / 2
        cands = [t + 10 * pipe for pipe in (0, 2) for t in (1, 2) if not (t == 1 and (a.out_dtype == F32 or a.Cout <= 64))]
        if a.dtype != F32 and a.out_dtype == a.dtype and a.Cout >= 128 and (a.Cin * 2) % 128 == 0:
            cands.append(28)                 # the 8-wavefront 128x128 tile carries the pre term as well
        if (a.dtype != F32 and a.out_dtype == a.dtype and (a.Cin * 2) % 128 == 0 and a.Cout % 8 == 0 and OPT.stream_gemm and a.act == ACT_SILU
                and (a.kh, a.kw, a.sh, a.sw, a.ph, a.pw) == (1, 1, 1, 1, 0, 0) and a.groups == 1):
            cands.append(52)                 # ... and so does the persistent streaming GEMM (1x1 SiLU layers)
            if a.Cout > 64:
                cands.append(51)
    for pipe in (() if (a.pre or a.w2) else CONV_PIPELINES):
        for t in (1, 2, 3, 4):
            if t == 1 and (a.out_dtype == F32 or a.Cout <= 64):
                continue
            if t == 3 and a.Cout > 32:
                continue
            cands.append(t + 10 * pipe)
    if a.dtype != F32 and a.out_dtype == a.dtype and not a.pre and not a.w2:            # 8-wavefront tiles (128-byte LDS-DMA pipeline only)
        if a.Cout >= 128:
            cands.append(25)
            cands.append(28)
        if a.Cout >= 256:
            cands.append(26)
    if (a.dtype != F32 and a.out_dtype == a.dtype and not a.pre and not a.w2 and (a.Cin * 2) % 128 == 0 and a.Cout % 8 == 0 and OPT.stream_gemm):
        cands.append(52)                     # persistent streaming implicit GEMM (igemm_stream.hip): 128 x 64 tile ...
        if a.Cout > 64:
            cands.append(51)                 # ... and 128 x 128; a launch the shape rules out returns an error and is skipped
    if cs_ok and not a.w2:
        cands.append(71)
    if cw and not a.w2:
        cands += cw
    if a.wf and OPT.wreg_gemm and not a.pre and not a.w2:
        cands.append(61)                     # weights fed from registers (igemm_wreg.hip): 128 x 128 ...
        if a.Cout > 128 and -(-a.Cout // 256) * 256 <= -(-a.Cout // 128) * 128:
            cands.append(62)                 # ... and 128 x 256 (its last channel tile must stay inside the packed Np = Cout rounded up to 128: wreg_check)
        # round 4: a wave owns 64 channels — the pixel feed per MAC halves
        if a.Cout > 128 and a.Cout % 256 == 0:
            cands.append(64)                 # 128 x 256 with four waves: two workgroups per CU
        if a.Cout > 256 and a.Cout % 512 == 0 and a.act != ACT_GELU:
            cands.append(63)                 # 128 x 512 with eight waves
        if a.B * a.Ho * a.Wo * a.groups <= OPT.wreg64_maxpix:  # few pixels (the 20 x 20 / 40 x 40 rows at batch 32): 64-pixel tiles double the workgroups
            cands.append(66)                 # 64 x 128, four waves x 32 channels
            if a.Cout > 128 and a.Cout % 256 == 0:
                cands.append(65)             # 64 x 256, four waves x 64 channels
    if (a.kh, a.kw, a.ph, a.pw) == (3, 3, 1, 1) and a.act == ACT_SILU and a.out_dtype == a.dtype and not a.pre and not a.w2:
        for shape, (bn, stride) in CTILE_SHAPES.items():       # 3x3 direct convolution from an LDS halo patch
            if a.sh == stride and a.sw == stride and a.Cout <= bn and (bn < 64 or a.Cout > bn // 2):
                cands.append(40 + shape)
    return cands


def autotune_conv(launch, stream_ptr, reps=3, context=()):
    """Time every (tile, pipeline) configuration of one recorded conv launch on the device and keep the fastest.
    All configurations walk K in the same order, so the result is bit-identical whichever is picked.
    context: the launches that precede this one in its plan — replayed (untimed) before every timed launch so that L2 /
    MALL hold what they hold in the real forward (the layer's input freshly written by its producers, not the layer's
    own previous run); timing a layer against itself back to back ranks the candidates wrongly at the margin."""
    a = launch.keep[0]
    sig = _conv_signature(a)
    cands = conv_candidates(a)
    keep = None
    if sig in _TUNE_CACHE:
        # A cached id is only as good as the state it was tuned under: the signature does not encode whether the fragment-major
        # weights exist (ICAF_WREG_GEMM / ICAF_CWIDE), which A/B switches are set, or the device's CU count (the persistent
        # streaming GEMM needs its channel tiles to divide an XCD's workgroups).  The entry is applied only if it is still a candidate
        # for THIS launch and the library's own check for that configuration accepts it; otherwise it is dropped and the launch re-tuned.
        if tile_valid(launch, _TUNE_CACHE[sig], cands):
            cached = _TUNE_CACHE[sig]
            fresh = [c for c in cands if (c in OPT.retune_tiles or (OPT.retune_pre and a.pre)) and c != cached]
            if not fresh or sig in _RETUNED:
                a.tile = cached
                return a.tile
            # ICAF_RETUNE_TILES: configurations added after the cache was written get their chance against the cached choice (and only
            # against it), and must beat it by 3 % — the per-launch timing mis-ranks by a few per cent from run to run
            _RETUNED.add(sig)
            keep, cands = cached, [cached] + fresh
        else:
            del _TUNE_CACHE[sig]
    best, best_ms = 0, float("inf")
    e0, e1 = Event(), Event()
    for c in cands:
        a.tile = c
        st = launch.fn(*launch.args, stream_ptr)
        if st != 0:
            continue
        if context:
            ms = 0.0
            for _ in range(reps):
                for l in context:
                    l(stream_ptr)
                e0.record(stream_ptr)
                launch.fn(*launch.args, stream_ptr)
                e1.record(stream_ptr)
                ms += e0.elapsed_ms(e1)
        else:
            e0.record(stream_ptr)
            for _ in range(reps):
                launch.fn(*launch.args, stream_ptr)
            e1.record(stream_ptr)
            ms = e0.elapsed_ms(e1)
        if keep is not None and c != keep:
            ms *= 1.03                          # a newcomer must win clearly
        if ms < best_ms:
            best, best_ms = c, ms
    a.tile = best
    _TUNE_CACHE[sig] = best
    return best


def tile_valid(launch, tile, cands=None):
    """Is launch configuration `tile` usable for this recorded conv launch — a candidate for its arguments AND accepted by the
    library's check of that configuration (icaf_conv2d_kernel_name runs the same *_check functions the launch runs)?"""
    a = launch.keep[0]
    if tile not in (conv_candidates(a) if cands is None else cands):
        return False
    saved = a.tile
    a.tile = tile
    try:
        buf = C.create_string_buffer(256)
        return lib().icaf_conv2d_kernel_name(launch.args[0], buf, 256) == 0
    finally:
        a.tile = saved


def save_tune_cache(path):
    import json
    with open(path, "w") as f:
        json.dump([[list(k), v] for k, v in _TUNE_CACHE.items()], f)


def load_tune_cache(path):
    import json
    with open(path) as f:
        for k, v in json.load(f):
            _TUNE_CACHE[tuple(k)] = v


def conv_kernel_name(launch):
    buf = C.create_string_buffer(256)
    check(lib().icaf_conv2d_kernel_name(launch.args[0], buf, 256), "icaf_conv2d_kernel_name")
    return buf.value.decode()


def preprocess(img, out, mode, name="preprocess"):
    """img: (B, C, H, W) fp32 NCHW contiguous; out: act (B, H', W', Cpad)."""
    assert img.dtype == torch.float32 and img.is_contiguous()
    if img.dim() == 5:                # both streams' images stacked (2, B, C, H, W): one launch over 2B images
        img, out = img.view(-1, *img.shape[2:]), flat_pair(out)
    B, Cc, H, W = img.shape
    Bo, Ho, Wo, cpad, ldo = _act_geom(out)
    assert ldo == cpad and Bo == B
    assert (Ho, Wo) == ((H // 2, W // 2) if mode == 1 else (H, W))
    nb = img.numel() * 4 + out.numel() * out.element_size()
    return Launch(lib().icaf_preprocess_nchw, (img.data_ptr(), out.data_ptr(), dtype_code(out.dtype), B, Cc, H, W,
                                               cpad, mode), keep=(img, out), name=name, nbytes=nb)


def preprocess_u8(img, out, mode, c0=0, name="preprocess_u8"):
    """img: (B, Ctot, H, W) uint8 NCHW contiguous (the dataloader's RGB+IR batch); out: act (B, H', W', Cpad) fed from
    channels [c0, c0+3), or a pair act (2, B, H', W', Cpad) fed from [c0, c0+3) and [c0+3, c0+6)."""
    assert img.dtype == torch.uint8 and img.is_contiguous() and img.dim() == 4
    nstreams = 2 if out.dim() == 5 else 1
    B, Ctot, H, W = img.shape
    o = flat_pair(out)
    Bo, Ho, Wo, cpad, ldo = _act_geom(o)
    assert ldo == cpad and Bo == nstreams * B and c0 + 3 * nstreams <= Ctot
    assert (Ho, Wo) == ((H // 2, W // 2) if mode == 1 else (H, W))
    nb = nstreams * B * 3 * H * W + o.numel() * o.element_size()
    return Launch(lib().icaf_preprocess_u8, (img.data_ptr(), o.data_ptr(), dtype_code(o.dtype), B, Ctot, c0, 3, nstreams,
                                             H, W, cpad, mode), keep=(img, out), name=name, nbytes=nb)


def stem(img, w_packed, kp, bias, y, cout, name="stem"):
    """Staging + 6x6/s2/p2 stem conv in one persistent kernel (icaf_stem).  img: fp32 (B, 3, H, W) / (2, B, 3, H, W)
    [both streams], or uint8 (B, 6, H, W) [both streams]; y: act or pair act of cout channels at half resolution."""
    u8 = img.dtype == torch.uint8
    paired = y.dim() == 5
    assert img.is_contiguous() and (u8 or img.dtype == torch.float32)
    if u8:
        assert img.dim() == 4 and paired and img.shape[1] >= 6
        B, ctot, H, W = img.shape
    else:
        assert img.dim() == (5 if paired else 4)
        B, _, H, W = img.shape[-4:]
        ctot = 3
    By, Ho, Wo, cy, ldy = _act_geom(y)
    assert (By, Ho, Wo) == (B, H // 2, W // 2) and cy >= cout and w_packed.dtype == y.dtype
    g = 2 if paired else 1
    nb = g * B * 3 * H * W * (1 if u8 else 4) + g * B * Ho * Wo * cout * y.element_size()
    flops = 2.0 * g * B * Ho * Wo * cout * 144
    return Launch(lib().icaf_stem, (img.data_ptr(), int(u8), ctot, w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), ldy,
                                    dtype_code(y.dtype), g, B, H, W, cout, kp,
                                    w_packed.stride(0) if paired else 0, bias.stride(0) if paired else 0,
                                    y.stride(0) if paired else 0),
                  keep=(img, w_packed, bias, y), name=name, flops=flops, nbytes=nb)


def stem2(img, w0, kp0, b0, w1, kp1, b1, w2, kp2, b2, y, c0, c1, c2, name="stem+conv3x3s2+1x1"):
    """Stem, the 3x3/s2 conv behind it and a chained 1x1 in one persistent kernel (icaf_stem2): img as `stem`; w0 / w1 /
    w2: the packed weights of the three layers (stacked per stream for a pair act y); y: (B, H/4, W/4, >= c2) act."""
    u8 = img.dtype == torch.uint8
    paired = y.dim() == 5
    assert img.is_contiguous() and (u8 or img.dtype == torch.float32)
    if u8:
        assert img.dim() == 4 and paired and img.shape[1] >= 6
        B, ctot, H, W = img.shape
    else:
        assert img.dim() == (5 if paired else 4)
        B, _, H, W = img.shape[-4:]
        ctot = 3
    By, Ho, Wo, cy, ldy = _act_geom(y)
    assert (By, Ho, Wo) == (B, (H // 2 - 1) // 2 + 1, (W // 2 - 1) // 2 + 1) and cy >= c2
    assert w0.dtype == w1.dtype == w2.dtype == y.dtype
    a = Stem2Args()
    a.img, a.img_u8, a.ctot = img.data_ptr(), int(u8), ctot
    a.dtype, a.nstreams, a.B, a.H, a.W = dtype_code(y.dtype), 2 if paired else 1, B, H, W
    for k, (w, b, kp, c) in enumerate(((w0, b0, kp0, c0), (w1, b1, kp1, c1), (w2, b2, kp2, c2))):
        setattr(a, f"w{k}", w.data_ptr()); setattr(a, f"bias{k}", b.data_ptr())
        setattr(a, f"w{k}_gs", w.stride(0) if paired else 0); setattr(a, f"bias{k}_gs", b.stride(0) if paired else 0)
        setattr(a, f"Kp{k}", kp); setattr(a, f"C{k}", c)
    a.y, a.y_gs, a.ldy = y.data_ptr(), y.stride(0) if paired else 0, ldy
    g = a.nstreams
    hs, ws = H // 2, W // 2
    flops = 2.0 * g * B * (hs * ws * c0 * 144 + Ho * Wo * (c1 * 9 * c0 + c2 * c1))
    nb = g * B * (3 * H * W * (1 if u8 else 4) + Ho * Wo * c2 * y.element_size())
    return Launch(lib().icaf_stem2, (C.byref(a),), keep=(a, img, w0, b0, w1, b1, w2, b2, y), name=name, flops=flops, nbytes=nb)


def sppf_pool(x, y1, y2, y3, k, name="sppf_pool"):
    x, y1, y2, y3 = flat_pair(x), flat_pair(y1), flat_pair(y2), flat_pair(y3)
    B, H, W, Cc, ldx = _act_geom(x)
    ldy = _act_geom(y1)[4]
    assert _act_geom(y2)[4] == ldy and _act_geom(y3)[4] == ldy
    nb = 4 * x.numel() * x.element_size()
    return Launch(lib().icaf_sppf_pool, (x.data_ptr(), ldx, y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), ldy,
                                         dtype_code(x.dtype), B, H, W, Cc, k), keep=(x, y1, y2, y3), name=name,
                  nbytes=nb)


def upsample_nearest(x, y, scale, name="upsample_nearest"):
    B, H, W, Cc, ldx = _act_geom(x)
    By, Hy, Wy, Cy, ldy = _act_geom(y)
    assert (Hy, Wy, Cy) == (H * scale, W * scale, Cc)
    nb = (x.numel() + y.numel()) * x.element_size()
    return Launch(lib().icaf_upsample_nearest, (x.data_ptr(), ldx, y.data_ptr(), ldy, dtype_code(x.dtype), B, H, W,
                                                Cc, scale), keep=(x, y), name=name, nbytes=nb)


def copy_channels(x, y, name="copy_channels"):
    B, H, W, Cc, ldx = _act_geom(x)
    ldy = _act_geom(y)[4]
    assert y.shape == x.shape
    nb = 2 * x.numel() * x.element_size()
    return Launch(lib().icaf_copy_channels, (x.data_ptr(), ldx, y.data_ptr(), ldy, dtype_code(x.dtype),
                                             B * H * W, Cc), keep=(x, y), name=name, nbytes=nb)


def axpby(x0, x1, y, a, b, name="add_fusion"):
    """y = a*x0 + b*x1 over acts of equal shape."""
    B, H, W, Cc, ld0 = _act_geom(x0)
    ld1, ldy = _act_geom(x1)[4], _act_geom(y)[4]
    assert x1.shape == x0.shape and y.shape == x0.shape
    nb = 3 * x0.numel() * x0.element_size()
    return Launch(lib().icaf_axpby, (x0.data_ptr(), ld0, x1.data_ptr(), ld1, y.data_ptr(), ldy, dtype_code(x0.dtype),
                                     B * H * W, Cc, float(a), float(b)), keep=(x0, x1, y), name=name, nbytes=nb)


def dmff_pool_tokens(fea_rgb, fea_ir, pos_rgb, pos_ir, tokens, th, tw, kh, kw, sh, sw, w_rgb, w_ir,
                     name="dmff_pool_tokens"):
    B, H, W, Cc, ld0 = _act_geom(fea_rgb)
    ld1 = _act_geom(fea_ir)[4]
    assert tokens.shape == (2, B * th * tw, Cc) and tokens.is_contiguous()
    assert pos_rgb.dtype == torch.float32 and pos_rgb.numel() == th * tw * Cc
    nb = 2 * fea_rgb.numel() * fea_rgb.element_size() + tokens.numel() * tokens.element_size()
    return Launch(lib().icaf_dmff_pool_tokens,
                  (fea_rgb.data_ptr(), ld0, fea_ir.data_ptr(), ld1, pos_rgb.data_ptr(), pos_ir.data_ptr(),
                   tokens.data_ptr(), dtype_code(tokens.dtype), B, H, W, Cc, th, tw, kh, kw, sh, sw,
                   float(w_rgb[0]), float(w_rgb[1]), float(w_ir[0]), float(w_ir[1])),
                  keep=(fea_rgb, fea_ir, pos_rgb, pos_ir, tokens), name=name, nbytes=nb)


def layernorm(x, y, g0, b0, g1, b1, eps=1e-5, name="layernorm"):
    """x, y: (G, rows, C) contiguous; group g normalised with (g_g, b_g) fp32 vectors."""
    G, rows, Cc = x.shape
    assert x.is_contiguous() and y.is_contiguous() and y.shape == x.shape
    nb = 2 * x.numel() * x.element_size()
    return Launch(lib().icaf_layernorm, (x.data_ptr(), y.data_ptr(), g0.data_ptr(), b0.data_ptr(), g1.data_ptr(),
                                         b1.data_ptr(), dtype_code(x.dtype), rows, Cc, G, float(eps)),
                  keep=(x, y, g0, b0, g1, b1), name=name, nbytes=nb)


def cross_attention(qkv, out, B, N, heads, name="cross_attention"):
    """qkv: (2, B*N, 3C); out: (2, B*N, C)."""
    G, rows, c3 = qkv.shape
    Cc = c3 // 3
    assert G == 2 and rows == B * N and out.shape == (2, rows, Cc) and qkv.is_contiguous() and out.is_contiguous()
    flops = 2 * B * heads * 4.0 * N * N * (Cc // heads)
    nb = (qkv.numel() + out.numel()) * qkv.element_size()
    return Launch(lib().icaf_cross_attention, (qkv.data_ptr(), out.data_ptr(), dtype_code(qkv.dtype), B, N, Cc,
                                               heads), keep=(qkv, out), name=name, flops=flops, nbytes=nb)


def dmff_fused_lds_bytes(C_, N, heads, dt):
    """LDS bytes one icaf_dmff_attn_mlp workgroup needs for this shape, or None when the fused block kernels do not cover it."""
    if dt not in (torch.bfloat16, torch.float16, torch.float32):      # (fp32: the parity instantiation, C <= 128)
        return None
    sz = C.c_size_t(0)
    check(lib().icaf_dmff_attn_mlp_lds_bytes(int(C_), int(N), int(heads), dtype_code(dt), C.byref(sz)), "dmff_attn_mlp_lds_bytes")
    return None if sz.value == C.c_size_t(-1).value or sz.value > 160 * 1024 else sz.value


def _dmff_args(x, qkv, y, packs, ln, coef, eps, B, N, heads):
    """icaf_dmff_args for one block iteration.  x: (2, B*N, C) contiguous tokens; qkv: (2, B*N, 3C); y: (2, B*N, C) view with
    any group / row stride; packs = dict(qkv=, out=, fc1=, fc2=) of (weights [2][Np][Kp], Kp, bias [2][Np]) stacks."""
    G, rows, Cc = x.shape
    assert G == 2 and rows == B * N and x.is_contiguous() and (qkv is None or (qkv.shape == (2, rows, 3 * Cc) and qkv.is_contiguous()))
    a = DmffArgs()
    a.x, a.qkv = x.data_ptr(), (qkv.data_ptr() if qkv is not None else None)
    a.x_gs = x.stride(0)
    if y is not None:
        assert y.shape == (2, rows, Cc) and y.stride(2) == 1 and y.dtype == x.dtype
        a.y, a.y_gs, a.ldy = y.data_ptr(), y.stride(0), y.stride(1)
    for name, key in (("wqkv", "qkv"), ("wo", "out"), ("w1", "fc1"), ("w2", "fc2")):
        w, kp, b = packs[key]
        assert w.dtype == x.dtype and w.dim() in (3, 5) and b.dim() == 2          # ([2][Np][Kp], or its fragment-major copy)
        setattr(a, name, w.data_ptr()); setattr(a, "b" + name[1:], b.data_ptr())
        setattr(a, name + "_gs", w.stride(0)); setattr(a, "b" + name[1:] + "_gs", b.stride(0))
    a.Kp, a.Kp4 = packs["qkv"][1], packs["fc2"][1]
    assert packs["out"][1] == a.Kp and packs["fc1"][1] == a.Kp
    a.hidden = coef["hidden"]
    a.ln_attn_gamma[0], a.ln_attn_gamma[1] = ln["a1w"].data_ptr(), ln["a2w"].data_ptr()
    a.ln_attn_beta[0], a.ln_attn_beta[1] = ln["a1b"].data_ptr(), ln["a2b"].data_ptr()
    a.ln_mlp_gamma, a.ln_mlp_beta = ln["mw"].data_ptr(), ln["mb"].data_ptr()
    a.dtype, a.B, a.N, a.C, a.heads = dtype_code(x.dtype), B, N, Cc, heads
    a.eps_attn, a.eps_mlp = float(eps[0]), float(eps[2])
    co = coef["co"]
    a.coef_res_attn[0], a.coef_res_attn[1] = co[0], co[2]
    a.coef_acc_attn[0], a.coef_acc_attn[1] = co[1], co[3]
    a.coef_res_mlp[0], a.coef_res_mlp[1] = co[4], co[6]
    a.coef_acc_mlp[0], a.coef_acc_mlp[1] = co[5], co[7]
    return a


def dmff_ln_qkv(x, qkv, packs, ln, coef, eps, B, N, heads, name="dmff_ln_qkv"):
    """LayerNorm + the six Linear(C, C) projections of CrossAttention as one launch (icaf_dmff_ln_qkv)."""
    a = _dmff_args(x, qkv, None, packs, ln, coef, eps, B, N, heads)
    rows, Cc = x.shape[1], x.shape[2]
    es = x.element_size()
    return Launch(lib().icaf_dmff_ln_qkv, (C.byref(a),), keep=(a, x, qkv, packs, ln), name=name, flops=2.0 * 2 * rows * Cc * 3 * Cc,
                  nbytes=2 * (rows * Cc * es + 3 * Cc * Cc * es + rows * 3 * Cc * es))


def dmff_attn_mlp(x, qkv, y, packs, ln, coef, eps, B, N, heads, name="dmff_attn_mlp"):
    """Crossed attention + out-projection + LayerNorm + MLP of one block iteration as one launch (icaf_dmff_attn_mlp)."""
    a = _dmff_args(x, qkv, y, packs, ln, coef, eps, B, N, heads)
    rows, Cc = x.shape[1], x.shape[2]
    es, hid = x.element_size(), coef["hidden"]
    flops = 2.0 * (B * heads * 4.0 * N * N * (Cc // heads)) + 2.0 * 2 * rows * (Cc * Cc + 2 * Cc * hid)
    nbytes = 2 * (rows * 3 * Cc * es + 2 * rows * Cc * es + (Cc * Cc + 2 * Cc * hid) * es)
    return Launch(lib().icaf_dmff_attn_mlp, (C.byref(a),), keep=(a, x, qkv, y, packs, ln), name=name, flops=flops, nbytes=nbytes)


def dmff_wide_ok(C_, hidden, dt):
    """The three-launch block kernels (dmff_wide.hip) cover this shape: C = 128 (four wavefronts, 128-channel passes) / 256 / 512 (eight,
    256-channel passes), 16-bit types, hidden a multiple of the pass width."""
    wpass = 128 if C_ == 128 else 256
    es = 4 if dt == torch.float32 else 2
    lds = 64 * (C_ * es + 16) + 64 * (wpass * es + 16) + 8 * 64 * 4 + hidden * 4
    if dt == torch.float32:                          # the parity instantiation: C = 128 only (same code, fp32 MFMAs, erff)
        return C_ == 128 and hidden % wpass == 0 and lds <= 160 * 1024
    return dt in (torch.bfloat16, torch.float16) and C_ in (128, 256, 512) and hidden % wpass == 0 and lds <= 160 * 1024


def _wide_packs(packs):
    """packs with fragment-major weight copies (frag_weights: cached per packed tensor) — what the dmff_wide kernels read"""
    return {k: ((frag_weights(v[0]), v[1], v[2]) if k in ("qkv", "out", "fc1", "fc2") else v) for k, v in packs.items()}


def dmff_wide_ln_qkv(x, qkv, packs, ln, coef, eps, B, N, heads, name="dmff_ln_qkv"):
    """LayerNorm + the six Linear(C, C) projections, wide levels (icaf_dmff_wide_ln_qkv)."""
    wp = _wide_packs(packs)
    a = _dmff_args(x, qkv, None, wp, ln, coef, eps, B, N, heads)
    a.reserved = OPT.dmff_qkv_npass                      # output-channel passes per workgroup: 0 = automatic (icaf.h)
    rows, Cc = x.shape[1], x.shape[2]
    es = x.element_size()
    return Launch(lib().icaf_dmff_wide_ln_qkv, (C.byref(a),), keep=(a, x, qkv, wp, packs, ln), name=name, flops=2.0 * 2 * rows * Cc * 3 * Cc,
                  nbytes=2 * (rows * Cc * es + 3 * Cc * Cc * es + rows * 3 * Cc * es))




def dmff_wide_ksplit(N, C_, hidden):
    """Hidden-column split of icaf_dmff_wide_proj_mlp_split for a level: 2 where a modality's weights (9 C^2 16-bit elements) overflow an
    XCD's 4 MB L2 AND the level has few tokens per image (N <= 128: P5 of yolov5s — 100 tokens, i.e. 100 tiles of 64 rows for 256 CUs at
    batch 32, 4.7 MB of weights per modality), else 1.  Deliberately a function of the LEVEL (N, C), never of the batch size: the split
    changes the fp32 association of the fc2 sum, and a shard of a batch must reproduce the same rows of the full batch bit for bit
    (tests/test_gpu_fullsize.py; yolov5l's P4 — C = 512, N = 256, one tile per CU at batch 32 — measured slower with the split anyway)."""
    if C_ < 256:
        return 1                                     # icaf_dmff_wide_proj_mlp_split is built for the 256-channel passes (C = 256 / 512) only
    if OPT.dmff_ksplit:
        return OPT.dmff_ksplit if hidden % (256 * OPT.dmff_ksplit) == 0 else 1
    return 2 if (9 * C_ * C_ * 2 > 3 * 2 ** 20 and N <= 128 and hidden % 512 == 0) else 1


def dmff_wide_proj_mlp(x, att, y, packs, ln, coef, eps, B, N, heads, name="dmff_proj_mlp", partial=None, ksplit=1, x32=None, y32=None):
    """Out-projection + LayerNorm + MLP of one block iteration as one launch behind cross_attention (icaf_dmff_wide_proj_mlp); with
    ksplit > 1 (partial: fp32 (ksplit, 2, rows, C) scratch) the hidden columns are split over ksplit workgroups per tile and a second
    small launch reduces the partial sums: returns the LIST [icaf_dmff_wide_proj_mlp_split, icaf_dmff_wide_reduce]."""
    rows, Cc = x.shape[1], x.shape[2]
    assert att.shape == (2, rows, Cc) and att.is_contiguous() and att.dtype == x.dtype
    wp = _wide_packs(packs)
    a = _dmff_args(x, None, y, wp, ln, coef, eps, B, N, heads)
    es, hid = x.element_size(), coef["hidden"]
    flops = 2.0 * 2 * rows * (Cc * Cc + 2 * Cc * hid)
    nbytes = 2 * (3 * rows * Cc * es + (Cc * Cc + 2 * Cc * hid) * es)
    if y32 is not None:                # fp32 residual stream across the block's iterations (icaf.h: icaf_dmff_args.x32 / y32)
        for t in (x32, y32):
            assert t is None or (t.dtype == torch.float32 and t.shape == (2, rows, Cc) and t.is_contiguous())
        a.y32 = y32.data_ptr()
        a.x32 = x32.data_ptr() if x32 is not None else None
        nbytes += (2 if x32 is not None else 1) * 2 * rows * Cc * 4
    else:
        assert x32 is None
    if ksplit > 1:
        assert partial is not None and partial.dtype == torch.float32 and partial.is_contiguous() and partial.shape == (ksplit, 2, rows, Cc)
        flops += 2.0 * 2 * rows * Cc * Cc * (ksplit - 1)                         # the repeated out-projection
        nbytes += partial.numel() * 4
        main = Launch(lib().icaf_dmff_wide_proj_mlp_split, (C.byref(a), att.data_ptr(), partial.data_ptr(), int(ksplit)),
                      keep=(a, x, att, y, wp, packs, ln, partial, x32, y32), name=name, flops=flops, nbytes=nbytes)
        red = Launch(lib().icaf_dmff_wide_reduce, (C.byref(a), partial.data_ptr(), int(ksplit)), keep=(a, y, partial, wp, packs, y32),
                     name=name + "_reduce", nbytes=partial.numel() * 4 + 2 * 2 * rows * Cc * es)
        return [main, red]
    return Launch(lib().icaf_dmff_wide_proj_mlp, (C.byref(a), att.data_ptr()), keep=(a, x, att, y, wp, packs, ln, x32, y32), name=name, flops=flops, nbytes=nbytes)


def dmff_upsample_merge(tokens, fea_rgb, fea_ir, out, th, tw, name="dmff_upsample_merge"):
    B, H, W, Cc, ld0 = _act_geom(fea_rgb)
    ld1 = _act_geom(fea_ir)[4]
    Bo, Ho, Wo, Co, ldo = _act_geom(out)
    assert (Bo, Ho, Wo, Co) == (B, H, W, 2 * Cc) and tokens.shape == (2, B * th * tw, Cc)
    nb = (2 * fea_rgb.numel() + out.numel()) * out.element_size()
    return Launch(lib().icaf_dmff_upsample_merge,
                  (tokens.data_ptr(), fea_rgb.data_ptr(), ld0, fea_ir.data_ptr(), ld1, out.data_ptr(), ldo,
                   dtype_code(out.dtype), B, H, W, Cc, th, tw), keep=(tokens, fea_rgb, fea_ir, out), name=name,
                  nbytes=nb)


def detect_decode(p, z, logits, raw, na, no, row_offset, stride, anchors_px, name="detect_decode"):
    """p: fp32 act (B, ny, nx, >=na*no); z: (B, rows_total, no); raw: (B, na, ny, nx, no)."""
    B, ny, nx, cp, ldp = _act_geom(p)
    assert p.dtype == torch.float32 and z.dtype == torch.float32 and z.is_contiguous()
    arr = (C.c_float * (2 * na))(*[float(v) for v in anchors_px])
    nb = p.numel() * 4 + 3 * B * na * ny * nx * no * 4
    return Launch(lib().icaf_detect_decode,
                  (p.data_ptr(), ldp, z.data_ptr(), logits.data_ptr() if logits is not None else None,
                   raw.data_ptr() if raw is not None else None, B, ny, nx, na, no, z.shape[1], row_offset,
                   float(stride), arr), keep=(p, z, logits, raw, arr), name=name, nbytes=nb)


def detect_conv_ok(x, na, no, cin):
    """Can icaf_detect_conv (Detect level in one launch) take this level?  16-bit maps, 3 anchors, no in {6, 8, 14}, Cin * 2 % 128 == 0."""
    return x.dtype in (torch.bfloat16, torch.float16) and na == 3 and no in (6, 8, 14) and (cin * 2) % 128 == 0 and x.dim() == 4


def detect_conv(x, w_packed, kp, bias, z, logits, raw, na, no, row_offset, stride, anchors_px, cin, name="detect_conv+decode"):
    """One Detect level in ONE launch (icaf_detect_conv): x (B, ny, nx, >= cin) 16-bit act -> z (B, rows_total, no), logits, raw; the
    1x1 output conv (packed weights / bias as for conv2d) runs as the persistent streaming GEMM with the decode as its epilogue."""
    B, ny, nx, cx, ldx = _act_geom(x)
    assert cx >= cin and z.dtype == torch.float32 and z.is_contiguous() and w_packed.dtype == x.dtype
    a = ConvArgs()
    a.x, a.w, a.bias = x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None
    a.y = z.data_ptr()                                   # (never written: icaf_conv_args wants a non-null y)
    a.groups = 1
    a.B, a.H, a.W, a.Cin, a.ldx = B, ny, nx, cin, ldx
    a.Ho, a.Wo, a.Cout, a.ldy = ny, nx, na * no, na * no
    a.kh = a.kw = a.sh = a.sw = 1
    a.Kp, a.act = kp, ACT_NONE
    a.dtype, a.out_dtype = dtype_code(x.dtype), F32
    a.alpha_acc[0] = a.alpha_acc[1] = 1.0
    arr = (C.c_float * (2 * na))(*[float(v) for v in anchors_px])
    m = B * ny * nx
    nb = m * cin * x.element_size() + 3 * m * na * no * 4
    return Launch(lib().icaf_detect_conv,
                  (C.byref(a), z.data_ptr(), logits.data_ptr() if logits is not None else None, raw.data_ptr() if raw is not None else None,
                   na, no, z.shape[1], row_offset, float(stride), arr), keep=(a, x, w_packed, bias, z, logits, raw, arr), name=name,
                  flops=2.0 * m * na * no * cin, nbytes=nb)


class NmsRunner:
    """Pre-allocated NMS launch for a fixed (B, rows, nc) — graph-capturable; results stay on the device."""

    def __init__(self, B, rows, nc, device, multi_label=False, max_det=300, want_keep=True, block=None):
        """want_keep=False skips the kept-index output (torchvision's return value; one more small launch) — the serving
        pipeline only needs the detection rows.  `block`: a flat fp32 tensor of B * max_det * 6 + B elements to use as the detection block
        (the pipeline lays the blocks of consecutive steps side by side so that ONE collective moves all of them)."""
        self.B, self.rows, self.nc, self.max_det = B, rows, nc, max_det
        self.multi_label = bool(multi_label) and nc > 1
        sz = C.c_size_t(0)
        check(lib().icaf_nms_workspace_bytes(B, rows, nc, int(self.multi_label), C.byref(sz)), "nms_workspace")
        self.ws = torch.empty((max(sz.value, 16),), dtype=torch.uint8, device=device)
        from .dist import detection_block            # det + count in ONE allocation: the block the all-gather sends as it is
        if block is None:
            self.block, self.det, self.count = detection_block(B, max_det, device)
        else:
            n = B * max_det * 6
            assert block.dtype == torch.float32 and block.is_contiguous() and block.numel() == n + B and block.device == torch.device(device)
            self.block, self.det, self.count = block, block[:n].view(B, max_det, 6), block[n:].view(torch.int32)
        self.keep = torch.zeros((B, max_det), dtype=torch.int32, device=device) if want_keep else None

    def launch(self, pred, conf_thres, iou_thres, agnostic=False, classes=None, max_nms=30000, max_wh=4096.0,
               stream_ptr=None):
        assert pred.dtype == torch.float32 and pred.is_contiguous() and pred.shape == (self.B, self.rows, 5 + self.nc)
        cls_arr, ncls = None, 0
        if classes is not None:
            ncls = len(classes)
            cls_arr = (C.c_int * max(ncls, 1))(*[int(c) for c in classes])
        st = lib().icaf_nms(pred.data_ptr(), self.B, self.rows, self.nc, float(conf_thres), float(iou_thres),
                            int(self.multi_label), int(bool(agnostic)), cls_arr, ncls, self.max_det, int(max_nms),
                            float(max_wh), self.det.data_ptr(), self.count.data_ptr(), self.keep.data_ptr() if self.keep is not None else None,
                            self.ws.data_ptr(), self.ws.numel(),
                            stream_ptr if stream_ptr is not None else current_stream_ptr())
        check(st, "icaf_nms")
        return self.det, self.count, self.keep


# ------------------------------------------------------------------------------------------------------------
# graph capture + events
# ------------------------------------------------------------------------------------------------------------
def match_predictions(det, count, labels, label_off, iouv, scale=None, predn=None, stream_ptr=None):
    """TP flags of every detection at every IoU threshold on the device (icaf_match_predictions; reference
    test.py:196-230).  det (B, max_det, 6) / count (B,) int32: the NMS output block; labels (L, 5) fp32 [cls, x1, y1,
    x2, y2] in native image space sorted by image, label_off (B + 1,) int32; scale (B, 5) fp32 [gain, pad_x, pad_y, w0,
    h0] or None; iouv (T,) fp32.  Returns uint8 (B, max_det, T); rows >= count[b] are unspecified."""
    B, max_det, six = det.shape
    assert six == 6 and det.dtype == torch.float32 and det.is_contiguous() and count.dtype == torch.int32
    assert label_off.dtype == torch.int32 and label_off.numel() == B + 1 and iouv.dtype == torch.float32
    assert labels.dtype == torch.float32 and labels.is_contiguous() and (labels.numel() == 0 or labels.shape[1] == 5)
    T = iouv.numel()
    correct = torch.empty((B, max_det, T), dtype=torch.uint8, device=det.device)
    off = label_off.cpu()
    max_l = int((off[1:] - off[:-1]).max()) if B else 0
    st = lib().icaf_match_predictions(det.data_ptr(), count.data_ptr(), B, max_det, labels.data_ptr() if labels.numel() else None,
                                      label_off.data_ptr(), max_l, scale.data_ptr() if scale is not None else None, iouv.data_ptr(), T,
                                      correct.data_ptr(), predn.data_ptr() if predn is not None else None,
                                      stream_ptr if stream_ptr is not None else current_stream_ptr())
    check(st, "icaf_match_predictions")
    return correct


class Graph:
    def __init__(self):
        self.exec = C.c_void_p(None)

    def capture(self, stream_ptr, fn):
        check(lib().icaf_graph_begin(stream_ptr), "graph_begin")
        try:
            fn()
        finally:
            st = lib().icaf_graph_end(stream_ptr, C.byref(self.exec))
        check(st, "graph_end")

    def launch(self, stream_ptr):
        check(lib().icaf_graph_launch(self.exec, stream_ptr), "graph_launch")

    def __del__(self):
        try:
            if self.exec:
                lib().icaf_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.ev = C.c_void_p(None)
        check(lib().icaf_event_create(C.byref(self.ev)), "event_create")

    def record(self, stream_ptr):
        check(lib().icaf_event_record(self.ev, stream_ptr), "event_record")

    def wait(self, stream_ptr):
        """Make `stream_ptr` wait for this event (inside a capture: adds a dependency edge / joins the capture)."""
        check(lib().icaf_stream_wait_event(stream_ptr, self.ev), "stream_wait_event")

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        check(lib().icaf_event_elapsed_ms(self.ev, stop.ev, C.byref(ms)), "event_elapsed")
        return ms.value

    def __del__(self):
        try:
            lib().icaf_event_destroy(self.ev)
        except Exception:
            pass


def device_info():
    cu, lds = C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(64)
    check(lib().icaf_device_info(C.byref(cu), C.byref(lds), buf, 64), "device_info")
    return {"cu_count": cu.value, "lds_bytes": lds.value, "arch": buf.value.decode()}
