"""Kernel-level parity on the MI355X: every C-ABI entry point vs the CPU oracle / plain fp32 torch on the same seeded
inputs.  fp32 builds are held to 1e-3 relative (they land around 1e-6); bf16 / f16 builds carry their own tolerance,
stated next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from icafusion_amd import ops            # noqa: E402
from icafusion_amd.engine import Plan    # noqa: E402
from oracle import icaf_oracle as oracle  # noqa: E402

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}   # relative to max |ref|


def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.normal(0, scale, shape)).astype(np.float32))


def to_act(x_nchw, dt, pad_to=None):
    """CPU NCHW fp32 -> GPU NHWC act (optionally a channel slice of a wider buffer to exercise ld > C)."""
    t = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV).to(dt)
    if pad_to:
        buf = torch.full((*t.shape[:3], pad_to), 7.0, dtype=dt, device=DEV)
        off = (pad_to - t.shape[3]) // 2 // 8 * 8
        buf[..., off:off + t.shape[3]] = t
        return buf[..., off:off + t.shape[3]]
    return t


def from_act(y):
    return y.float().cpu().permute(0, 3, 1, 2)


def run(launch):
    launch(ops.current_stream_ptr())
    torch.cuda.synchronize()


def close(got, ref, dt, what="", factor=1.0):
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max()) / scale
    assert err <= TOL[dt] * factor, f"{what}: rel err {err:.3e} > {TOL[dt] * factor:.1e} ({dt})"


def q(x, dt):
    """Round-trip through the storage dtype (what the kernel actually sees)."""
    return x.to(dt).float()


CONV_CASES = [
    # B, H, W, cin, cout, k, s, p, act, res, tile
    (2, 20, 24, 32, 64, 1, 1, 0, ops.ACT_SILU, False, 0),
    (1, 33, 17, 64, 96, 3, 1, 1, ops.ACT_SILU, True, 0),
    (2, 32, 32, 32, 64, 3, 2, 1, ops.ACT_SILU, False, 0),
    (1, 16, 16, 16, 32, 3, 1, 1, ops.ACT_SILU, False, 3),      # Cin < BK: two taps per K slice
    (1, 40, 40, 128, 256, 1, 1, 0, ops.ACT_NONE, False, 1),
    (1, 40, 40, 128, 256, 1, 1, 0, ops.ACT_GELU, True, 2),
    (3, 10, 10, 256, 128, 3, 1, 1, ops.ACT_SILU, True, 4),
    (1, 13, 13, 72, 40, 3, 2, 1, ops.ACT_SILU, False, 0),      # ragged channel counts (not multiples of the tile)
    (1, 8, 8, 512, 18, 1, 1, 0, ops.ACT_NONE, False, 0),       # Detect-like head
    # register-staged fallback pipeline (tile ids 11..14), kept for operands beyond the 2 GiB descriptor range
    (1, 33, 17, 64, 96, 3, 1, 1, ops.ACT_SILU, True, 11),
    (2, 32, 32, 32, 64, 3, 2, 1, ops.ACT_SILU, False, 12),
    (1, 16, 16, 16, 32, 3, 1, 1, ops.ACT_GELU, False, 13),
    (3, 10, 10, 256, 128, 3, 1, 1, ops.ACT_NONE, True, 14),
    (2, 23, 29, 48, 160, 3, 1, 1, ops.ACT_SILU, True, 2),      # M, N, K all ragged on the DMA pipeline
    (1, 9, 7, 32, 32, 5, 2, 2, ops.ACT_SILU, False, 4),        # 5x5 stride 2
    # 128-byte-slice DMA pipelines (2-stage ring = 2x, 3-stage ring = 3x)
    (1, 33, 17, 64, 96, 3, 1, 1, ops.ACT_SILU, True, 21),
    (2, 32, 32, 32, 64, 3, 2, 1, ops.ACT_SILU, False, 22),
    (1, 16, 16, 16, 32, 3, 1, 1, ops.ACT_SILU, False, 23),
    (3, 10, 10, 256, 128, 3, 1, 1, ops.ACT_GELU, True, 24),
    (1, 40, 40, 128, 256, 1, 1, 0, ops.ACT_NONE, True, 31),
    (2, 23, 29, 48, 160, 3, 1, 1, ops.ACT_SILU, True, 32),
    (1, 20, 20, 24, 32, 3, 1, 1, ops.ACT_SILU, False, 33),
    (1, 13, 13, 72, 40, 3, 2, 1, ops.ACT_SILU, False, 34),
    # 8-wavefront 128-row tiles (ids 28 / 29)
    (2, 20, 20, 128, 256, 3, 1, 1, ops.ACT_SILU, True, 28),
    (1, 23, 29, 64, 136, 3, 2, 1, ops.ACT_GELU, True, 28),
    (1, 33, 17, 64, 96, 3, 1, 1, ops.ACT_SILU, True, 29),
    (2, 40, 40, 128, 40, 1, 1, 0, ops.ACT_NONE, False, 29),
    # 8-wavefront tiles 256x128 / 256x256 (16-bit types; the fp32 run of these rows falls back to tile 2)
    (2, 20, 20, 128, 256, 3, 1, 1, ops.ACT_SILU, True, 25),
    (1, 40, 40, 64, 136, 1, 1, 0, ops.ACT_GELU, True, 25),      # ragged N on the 256x128 tile
    (2, 20, 20, 256, 512, 1, 1, 0, ops.ACT_SILU, False, 26),
    (1, 23, 29, 128, 384, 3, 2, 1, ops.ACT_NONE, True, 26),     # N = 1.5 tiles: rows past the packed weights read as zero
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, dt):
    B, H, W, cin, cout, k, s, p, act, use_res, tile = case
    if tile % 10 == 1 and dt == torch.float32:
        tile += 1
    if tile in (25, 26, 28, 29) and dt == torch.float32:
        tile = 2
    x = rnd((B, cin, H, W), 1)
    w = rnd((cout, cin, k, k), 2, 1.0 / math.sqrt(cin * k * k))
    bias = rnd((cout,), 3, 0.2)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd((B, cout, Ho, Wo), 4) if use_res else None
    xa = to_act(x, dt, pad_to=cin + 16)
    wp, kp = ops.pack_conv_weight(w.to(DEV), dt)
    bp = ops.pack_bias(bias.to(DEV), cout)
    vec = ops.VEC[dt]
    ldy = -(-cout // vec) * vec + vec
    ybuf = torch.zeros((B, Ho, Wo, ldy), dtype=dt, device=DEV)
    y = ybuf[..., :cout]
    ra = to_act(res, dt) if use_res else None
    run(ops.conv2d(xa, wp, kp, bp, y, k, k, s, s, p, p, cin, cout, act, res=ra, alpha_acc=0.75, alpha_res=1.25, tile=tile))
    ref = F.conv2d(q(x, dt), q(w, dt), bias, s, p)
    ref = {ops.ACT_NONE: lambda t: t, ops.ACT_SILU: F.silu, ops.ACT_GELU: F.gelu}[act](ref) * 0.75
    if use_res:
        ref = ref + 1.25 * q(res, dt)
    close(from_act(y), ref, dt, f"conv {case}")
    assert float(ybuf[..., cout:].abs().max()) == 0.0, "conv wrote outside its channel slice"


STREAM_CASES = [
    # B, H, W, cin, cout, k, stride, act, use_res, groups, shape (tile id 50 + shape): persistent streaming implicit GEMM (igemm_stream.hip)
    (8, 80, 80, 128, 128, 1, 1, ops.ACT_SILU, False, 2, 1),     # C3 cv3 at 80x80, both backbones side by side: 6 tiles per workgroup
    (8, 80, 80, 64, 64, 1, 1, ops.ACT_SILU, False, 2, 2),       # ONE 128-byte slice per tile, 128 x 64 tile
    (12, 37, 41, 256, 256, 1, 1, ops.ACT_SILU, False, 1, 1),    # two channel tiles per pixel tile, 143 pixel tiles (1-2 per workgroup), ragged last one
    (4, 40, 40, 512, 256, 1, 1, ops.ACT_NONE, True, 1, 1),      # K = 8 slices, residual (DMFF fc2-like mix)
    (2, 64, 72, 128, 512, 1, 1, ops.ACT_GELU, False, 2, 1),     # four channel tiles, GELU (DMFF fc1)
    (5, 33, 29, 192, 64, 1, 1, ops.ACT_SILU, False, 1, 2),      # K = 3 slices (odd), Cout = one 64-wide tile
    (9, 40, 40, 1024, 512, 1, 1, ops.ACT_SILU, False, 1, 1),    # SPPF cv2: K = 16 slices
    (6, 52, 52, 64, 136, 1, 1, ops.ACT_NONE, False, 1, 2),      # three channel tiles do not divide an XCD's workgroups: rejected (checked below)
    (8, 80, 80, 64, 64, 3, 1, ops.ACT_SILU, True, 2, 2),        # Bottleneck 3x3 + shortcut at 80x80, paired: the tap walk restarts per tile
    (6, 41, 37, 128, 128, 3, 1, ops.ACT_SILU, False, 1, 1),     # 3x3, ragged map / last tile: taps leave the image on every side
    (4, 80, 80, 64, 128, 3, 2, ops.ACT_SILU, False, 2, 1),      # stride 2 down-sampling conv
    (3, 23, 31, 64, 64, 5, 2, ops.ACT_SILU, True, 1, 2),        # 5x5 / stride 2 / pad 2
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", STREAM_CASES)
def test_conv_streaming_kernel(case, dt):
    """igemm_stream.hip vs torch, and BIT-EXACT vs the implicit-GEMM kernel (same K order, MFMA step and epilogue expressions):
    a persistent workgroup walks several tiles through one DMA ring, so the cases cover several tiles per workgroup, ragged last
    tiles, one to sixteen K slices per tile, one to four channel tiles, 1x1 / 3x3 / 5x5 filters with stride and padding, residuals
    and the paired (groups = 2) launch."""
    B, H, W, cin, cout, k, st, act, use_res, G, shape = case
    p_ = k // 2
    Ho, Wo = (H + 2 * p_ - k) // st + 1, (W + 2 * p_ - k) // st + 1
    xs = [rnd((B, cin, H, W), 51 + g) for g in range(G)]
    ws = [rnd((cout, cin, k, k), 53 + g, 1.0 / math.sqrt(cin * k * k)) for g in range(G)]
    bs = [rnd((cout,), 55 + g, 0.2) for g in range(G)]
    rs = [rnd((B, cout, Ho, Wo), 57 + g) for g in range(G)] if use_res else None
    stk = (lambda t: torch.stack(t).contiguous()) if G == 2 else (lambda t: t[0])
    xa = stk([to_act(x, dt, pad_to=cin + 16) for x in xs])
    packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
    wp, kp = stk([p0[0] for p0 in packs]), packs[0][1]
    bp = stk([ops.pack_bias(b.to(DEV), cout) for b in bs])
    ra = stk([to_act(r, dt) for r in rs]) if use_res else None
    ldy = -(-cout // 8) * 8 + 8
    outs = []
    for tile in (50 + shape, 2):
        ybuf = torch.full((G, B, Ho, Wo, ldy) if G == 2 else (B, Ho, Wo, ldy), 7.0, dtype=dt, device=DEV)
        y = ybuf[..., :cout]
        try:
            run(ops.conv2d(xa, wp, kp, bp, y, k, k, st, st, p_, p_, cin, cout, act, res=ra, alpha_acc=0.75, alpha_res=1.25, tile=tile))
        except ops._lib.IcafError as e:
            assert tile > 50 and "channel tiles do not divide" in str(e) and cout == 136, e
            return
        assert float((ybuf[..., cout:] - 7.0).abs().max()) == 0.0, "conv wrote outside its channel slice"
        outs.append(y.clone())
    assert torch.equal(outs[0], outs[1]), f"streaming kernel != igemm, max diff {(outs[0].float() - outs[1].float()).abs().max().item()}"
    for g in range(G):
        ref = F.conv2d(q(xs[g], dt), q(ws[g], dt), bs[g], st, p_)
        ref = {ops.ACT_NONE: lambda t: t, ops.ACT_SILU: F.silu, ops.ACT_GELU: F.gelu}[act](ref) * 0.75
        if use_res:
            ref = ref + 1.25 * q(rs[g], dt)
        close(from_act(outs[0][g] if G == 2 else outs[0]), ref, dt, f"stream {case} group {g}")


def test_conv_streaming_kernel_rejects_other_layers():
    x = torch.zeros((1, 32, 32, 32), dtype=torch.bfloat16, device=DEV)          # Cin * 2 bytes = 64: a K slice would straddle taps
    w3 = rnd((64, 32, 3, 3), 1)
    wp, kp = ops.pack_conv_weight(w3.to(DEV), torch.bfloat16)
    y = torch.zeros((1, 32, 32, 64), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ops._lib.IcafError):
        run(ops.conv2d(x, wp, kp, None, y, 3, 3, 1, 1, 1, 1, 32, 64, ops.ACT_SILU, tile=52))


CSTREAM_CASES = [
    # B, H, W, cout, use_res, groups, chain (None | (cout2, keep)): persistent 3x3 with the filter resident in LDS (cstream.hip, tile id 71)
    (8, 80, 80, 64, True, 2, None),            # Bottleneck.cv2 + shortcut at 80 x 80, both backbones: 12-13 tiles per workgroup
    (8, 80, 80, 64, False, 1, None),           # head C3 (no shortcut)
    (3, 37, 45, 64, True, 1, None),            # ragged map: partial tiles on the right / bottom edge, fewer tiles than workgroups
    (2, 24, 16, 48, False, 1, None),           # Cout < 64
    (8, 80, 80, 64, True, 2, (64, True)),      # 3x3 + shortcut, then the next Bottleneck's 1x1 (chain_keep): y and y2
    (5, 41, 33, 64, False, 1, (64, False)),    # chained 1x1 without keeping y, ragged map
    (4, 40, 56, 64, True, 1, (32, True)),      # narrower chained layer
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CSTREAM_CASES)
def test_conv3x3_resident_filter_kernel(case, dt):
    """cstream.hip vs torch and BIT-EXACT vs igemm (same K order, MFMA step, epilogue / chain expressions).  The kernel pipelines a
    tile's stores into the next tile and double-buffers the halo patch: the cases run many tiles per workgroup, ragged maps, the
    residual (deferred and immediate forms), the chained 1x1 with and without chain_keep, and the paired launch."""
    B, H, W, cout, use_res, G, chain = case
    cin = 64
    xs = [rnd((B, cin, H, W), 71 + g) for g in range(G)]
    ws = [rnd((cout, cin, 3, 3), 73 + g, 1.0 / math.sqrt(cin * 9)) for g in range(G)]
    bs = [rnd((cout,), 75 + g, 0.2) for g in range(G)]
    rs = [rnd((B, cout, H, W), 77 + g) for g in range(G)] if use_res else None
    stk = (lambda t: torch.stack(t).contiguous()) if G == 2 else (lambda t: t[0])
    xa = stk([to_act(x, dt) for x in xs])
    packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
    wp, kp = stk([p0[0] for p0 in packs]), packs[0][1]
    bp = stk([ops.pack_bias(b.to(DEV), cout) for b in bs])
    ra = stk([to_act(r, dt) for r in rs]) if use_res else None
    ch = None
    if chain:
        c2, keep = chain
        w2s = [rnd((c2, cout, 1, 1), 81 + g, 1.0 / math.sqrt(cout)) for g in range(G)]
        b2s = [rnd((c2,), 83 + g, 0.2) for g in range(G)]
        p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2s]
        w2p, kp2 = stk([p0[0] for p0 in p2]), p2[0][1]
        b2p = stk([ops.pack_bias(b.to(DEV), c2) for b in b2s])
    outs = []
    for tile in (71, 2 if not chain else 22):
        shape = (G, B, H, W) if G == 2 else (B, H, W)
        y = torch.full(shape + (cout + 8,), 7.0, dtype=dt, device=DEV)[..., :cout]
        if chain:
            y2 = torch.full(shape + (c2 + 8,), 7.0, dtype=dt, device=DEV)[..., :c2]
            ch = dict(w=w2p, kp=kp2, bias=b2p, y=y2, cout=c2, keep=keep)
        run(ops.conv2d(xa, wp, kp, bp, y, 3, 3, 1, 1, 1, 1, cin, cout, ops.ACT_SILU, res=ra if (not chain or chain[1]) else None,
                       alpha_res=1.25 if not chain else 1.0, tile=tile, chain=ch))
        outs.append((y.clone(), ch["y"].clone() if chain else None))
    if not chain or chain[1]:
        assert torch.equal(outs[0][0], outs[1][0]), f"cstream y != igemm, max diff {(outs[0][0].float() - outs[1][0].float()).abs().max().item()}"
    if chain:
        assert torch.equal(outs[0][1], outs[1][1]), f"cstream y2 != igemm, max diff {(outs[0][1].float() - outs[1][1].float()).abs().max().item()}"
    for g in range(G):
        m = F.silu(F.conv2d(q(xs[g], dt), q(ws[g], dt), bs[g], 1, 1))
        if use_res and (not chain or chain[1]):
            m = m + (1.25 if not chain else 1.0) * q(rs[g], dt)
        if not chain or chain[1]:
            close(from_act(outs[0][0][g] if G == 2 else outs[0][0]), m, dt, f"cstream {case} group {g}", factor=2.0)
        if chain:
            close(from_act(outs[0][1][g] if G == 2 else outs[0][1]), F.silu(F.conv2d(q(m, dt), q(w2s[g], dt), b2s[g])), dt, f"cstream chain {case}", factor=2.0)


CWIDE_CASES = [
    # B, H, W, use_res, groups, chain (None | (cout2, keep)): 3x3 128 -> 128 from a resident halo patch, weights streamed into registers (cwide.hip, tile id 81)
    (8, 40, 40, True, 2, None),             # Bottleneck.cv2 + shortcut at 40 x 40, both backbones: 15 tiles per image (right column half empty)
    (64, 40, 40, True, 2, None),            # the bench's batch: the persistent form walks 3-4 tiles per workgroup
    (4, 40, 40, False, 1, None),            # head C3 (no shortcut)
    (3, 21, 27, True, 1, None),             # ragged map: partial tiles on both edges
    (1, 8, 16, False, 1, None),             # ONE tile
    (8, 40, 40, True, 2, (128, True)),      # 3x3 + shortcut, then the next Bottleneck's 1x1 (chain_keep): y and y2
    (3, 33, 17, False, 1, (128, False)),    # chained 1x1 without keeping y, ragged map
    (2, 24, 40, True, 1, (64, True)),       # narrower chained layer
]


CWIDE_PARAMS = [pytest.param(c, t, id=f"{i}-{n}") for i, c in enumerate(CWIDE_CASES) for t, n in ((81, "8x16"), (82, "8x8"))]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case,shape", CWIDE_PARAMS)
def test_conv3x3_resident_patch_streamed_weights_kernel(case, dt, shape):
    """cwide.hip vs torch and BIT-EXACT vs igemm (same K order, MFMA step, epilogue / chain expressions): ragged maps, the residual,
    the chained 1x1 with and without chain_keep, and the paired launch."""
    B, H, W, use_res, G, chain = case
    cin = cout = 128
    xs = [rnd((B, cin, H, W), 171 + g) for g in range(G)]
    ws = [rnd((cout, cin, 3, 3), 173 + g, 1.0 / math.sqrt(cin * 9)) for g in range(G)]
    bs = [rnd((cout,), 175 + g, 0.2) for g in range(G)]
    rs = [rnd((B, cout, H, W), 177 + g) for g in range(G)] if use_res else None
    stk = (lambda t: torch.stack(t).contiguous()) if G == 2 else (lambda t: t[0])
    xa = stk([to_act(x, dt) for x in xs])
    packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
    wp, kp = stk([p0[0] for p0 in packs]), packs[0][1]
    bp = stk([ops.pack_bias(b.to(DEV), cout) for b in bs])
    ra = stk([to_act(r, dt) for r in rs]) if use_res else None
    ch = None
    if chain:
        c2, keep = chain
        w2s = [rnd((c2, cout, 1, 1), 181 + g, 1.0 / math.sqrt(cout)) for g in range(G)]
        b2s = [rnd((c2,), 183 + g, 0.2) for g in range(G)]
        p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2s]
        w2p, kp2 = stk([p0[0] for p0 in p2]), p2[0][1]
        b2p = stk([ops.pack_bias(b.to(DEV), c2) for b in b2s])
    outs = []
    for tile in (shape, 28 if not chain else 21):
        shape = (G, B, H, W) if G == 2 else (B, H, W)
        y = torch.full(shape + (cout + 8,), 7.0, dtype=dt, device=DEV)[..., :cout]
        if chain:
            y2 = torch.full(shape + (c2 + 8,), 7.0, dtype=dt, device=DEV)[..., :c2]
            ch = dict(w=w2p, kp=kp2, bias=b2p, y=y2, cout=c2, keep=keep)
        run(ops.conv2d(xa, wp, kp, bp, y, 3, 3, 1, 1, 1, 1, cin, cout, ops.ACT_SILU, res=ra if (not chain or chain[1]) else None,
                       alpha_res=1.25 if not chain else 1.0, tile=tile, chain=ch))
        outs.append((y.clone(), ch["y"].clone() if chain else None))
    if not chain or chain[1]:
        assert torch.equal(outs[0][0], outs[1][0]), f"cwide y != igemm, max diff {(outs[0][0].float() - outs[1][0].float()).abs().max().item()}"
    if chain:
        assert torch.equal(outs[0][1], outs[1][1]), f"cwide y2 != igemm, max diff {(outs[0][1].float() - outs[1][1].float()).abs().max().item()}"
    for g in range(G):
        m = F.silu(F.conv2d(q(xs[g], dt), q(ws[g], dt), bs[g], 1, 1))
        if use_res and (not chain or chain[1]):
            m = m + (1.25 if not chain else 1.0) * q(rs[g], dt)
        if not chain or chain[1]:
            close(from_act(outs[0][0][g] if G == 2 else outs[0][0]), m, dt, f"cwide {case} group {g}", factor=2.0)
        if chain:
            close(from_act(outs[0][1][g] if G == 2 else outs[0][1]), F.silu(F.conv2d(q(m, dt), q(w2s[g], dt), b2s[g])), dt, f"cwide chain {case}", factor=2.0)


CWIDE_TAIL_CASES = [
    # B, H, W, use_res, groups: the C3 tail — 3x3 128 -> 128 (+ shortcut) carrying cv3 over cat(m, x2) -> 256 channels (icaf_conv_args.x2)
    (8, 40, 40, True, 2),        # backbone C3 at 40 x 40, both streams
    (4, 40, 40, False, 1),       # head C3 (no shortcut)
    (3, 21, 27, True, 1),        # ragged map: partial tiles on both edges
    (1, 8, 16, False, 1),        # ONE tile
    (32, 40, 40, True, 2),       # the bench's batch
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [81, 82])
@pytest.mark.parametrize("case", CWIDE_TAIL_CASES)
def test_conv3x3_with_c3_tail_equals_two_launches(case, dt, tile):
    """cwide.hip, CHAIN = 2: y2 = SiLU(W3 . cat([res +] SiLU(conv3x3(x)), x2) + b3) in one launch vs the 3x3 launch (igemm) followed by
    the 1x1 launch over the materialised concatenation (igemm): BIT-EXACT (same K order [m | x2], MFMA step and epilogue expressions),
    and close to torch on the rounded operands.  The intermediate m is not written (its buffer keeps the fill value)."""
    B, H, W, use_res, G = case
    c = 128
    xs = [rnd((B, c, H, W), 271 + g) for g in range(G)]
    ws = [rnd((c, c, 3, 3), 273 + g, 1.0 / math.sqrt(c * 9)) for g in range(G)]
    bs = [rnd((c,), 275 + g, 0.2) for g in range(G)]
    rs = [rnd((B, c, H, W), 277 + g) for g in range(G)] if use_res else None
    x2s = [rnd((B, c, H, W), 279 + g) for g in range(G)]
    w3s = [rnd((2 * c, 2 * c, 1, 1), 281 + g, 1.0 / math.sqrt(2 * c)) for g in range(G)]
    b3s = [rnd((2 * c,), 283 + g, 0.2) for g in range(G)]
    stk = (lambda t: torch.stack(t).contiguous()) if G == 2 else (lambda t: t[0])
    shape = (G, B, H, W) if G == 2 else (B, H, W)
    xa = stk([to_act(x, dt) for x in xs])
    packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
    wp, kp = stk([p0[0] for p0 in packs]), packs[0][1]
    bp = stk([ops.pack_bias(b.to(DEV), c) for b in bs])
    ra = stk([to_act(r, dt) for r in rs]) if use_res else None
    # the buffer cv3 reads in the two-launch form: [m | x2] (as C3.emit places them); the tail reads its x2 half in place
    cat = torch.full(shape + (2 * c,), 7.0, dtype=dt, device=DEV)
    for g in range(G):
        (cat[g] if G == 2 else cat)[..., c:] = to_act(x2s[g], dt)
    p3 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w3s]
    w3p, kp3 = stk([p0[0] for p0 in p3]), p3[0][1]
    b3p = stk([ops.pack_bias(b.to(DEV), 2 * c) for b in b3s])
    assert kp3 == 2 * c
    # one launch
    y_unused = torch.full(shape + (c,), 7.0, dtype=dt, device=DEV)
    y2 = torch.full(shape + (2 * c + 8,), 7.0, dtype=dt, device=DEV)[..., :2 * c]
    run(ops.conv2d(xa, wp, kp, bp, y_unused, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, res=ra, tile=tile,
                   chain=dict(w=w3p, kp=kp3, bias=b3p, y=y2, cout=2 * c, x2=cat[..., c:])))
    assert bool((y_unused == 7.0).all()), "the intermediate tensor must not be written"
    # two launches through igemm
    run(ops.conv2d(xa, wp, kp, bp, cat[..., :c], 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, res=ra, tile=28))
    y2_ref = torch.full(shape + (2 * c,), 7.0, dtype=dt, device=DEV)
    run(ops.conv2d(cat, w3p, kp3, b3p, y2_ref, 1, 1, 1, 1, 0, 0, 2 * c, 2 * c, ops.ACT_SILU, tile=21))
    assert torch.equal(y2, y2_ref), f"tail != two launches, max diff {(y2.float() - y2_ref.float()).abs().max().item()}"
    for g in range(G):
        m = F.silu(F.conv2d(q(xs[g], dt), q(ws[g], dt), bs[g], 1, 1))
        if use_res:
            m = m + q(rs[g], dt)
        ref = F.silu(F.conv2d(torch.cat((q(m, dt), q(x2s[g], dt)), 1), q(w3s[g], dt), b3s[g]))
        close(from_act(y2[g] if G == 2 else y2), ref, dt, f"cwide tail {case} group {g}", factor=2.0)


def test_c3_tail_argument_checks():
    """x2 without w2, on a launch configuration that is not built for it, or with a cv3 width other than 256: errors with a reason."""
    from icafusion_amd._lib import IcafError
    dt, c = torch.bfloat16, 128
    x = torch.zeros((1, 8, 16, c), dtype=dt, device=DEV)
    wp, kp = ops.pack_conv_weight(torch.zeros((c, c, 3, 3), device=DEV), dt)
    w3p, kp3 = ops.pack_conv_weight(torch.zeros((2 * c, 2 * c, 1, 1), device=DEV), dt)
    y, y2, x2 = torch.zeros_like(x), torch.zeros((1, 8, 16, 2 * c), dtype=dt, device=DEV), torch.zeros_like(x)
    for tile in (21, 28, 83):
        with pytest.raises(IcafError, match="x2"):
            run(ops.conv2d(x, wp, kp, None, y, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, tile=tile, chain=dict(w=w3p, kp=kp3, bias=None, y=y2, cout=2 * c, x2=x2)))
    with pytest.raises(IcafError, match="unknown launch configuration"):        # 90 + shape (the persistent forms) were removed in round 6: an error, never a silent pick
        run(ops.conv2d(x, wp, kp, None, y, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, tile=91))
    w1p, kp1 = ops.pack_conv_weight(torch.zeros((c, 2 * c, 1, 1), device=DEV), dt)
    with pytest.raises(IcafError, match="256"):
        run(ops.conv2d(x, wp, kp, None, y, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, tile=81, chain=dict(w=w1p, kp=kp1, bias=None, y=y2[..., :c], cout=c, x2=x2)))


CWIDE_S2_CASES = [
    # B, H, W, cin, cout, groups, chain cout2 (0 = none), tile id: stride-2 3x3 from a resident halo patch (even | odd column planes)
    (4, 160, 160, 64, 128, 2, 128, 83),      # backbone 160 -> 80 + the C3's cv1 | cv2 (chained 1x1), both backbones
    (2, 160, 160, 64, 128, 1, 128, 85),      # the same through the 8 x 8 tiling
    (3, 63, 75, 64, 128, 1, 0, 83),          # odd input size: ragged output map (32 x 38), partial tiles, bottom / right padding rows
    (2, 50, 34, 64, 256, 1, 0, 85),          # two channel blocks
    (4, 80, 80, 128, 256, 2, 0, 84),         # backbone 80 -> 40, both backbones, two channel blocks
    (3, 80, 80, 128, 128, 1, 0, 84),         # head down-sampling 80 -> 40
    (2, 37, 45, 128, 128, 1, 64, 84),        # ragged, chained narrower 1x1
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CWIDE_S2_CASES)
def test_conv3x3_stride2_resident_patch_kernel(case, dt):
    """cwide.hip, stride 2: vs torch and BIT-EXACT vs igemm."""
    B, H, W, cin, cout, G, c2, tile_id = case
    xs = [rnd((B, cin, H, W), 271 + g) for g in range(G)]
    ws = [rnd((cout, cin, 3, 3), 273 + g, 1.0 / math.sqrt(cin * 9)) for g in range(G)]
    bs = [rnd((cout,), 275 + g, 0.2) for g in range(G)]
    stk = (lambda t: torch.stack(t).contiguous()) if G == 2 else (lambda t: t[0])
    xa = stk([to_act(x, dt) for x in xs])
    packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
    wp, kp = stk([p0[0] for p0 in packs]), packs[0][1]
    bp = stk([ops.pack_bias(b.to(DEV), cout) for b in bs])
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if c2:
        w2s = [rnd((c2, cout, 1, 1), 281 + g, 1.0 / math.sqrt(cout)) for g in range(G)]
        b2s = [rnd((c2,), 283 + g, 0.2) for g in range(G)]
        p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2s]
        w2p, kp2 = stk([p0[0] for p0 in p2]), p2[0][1]
        b2p = stk([ops.pack_bias(b.to(DEV), c2) for b in b2s])
    outs = []
    for tile in (tile_id, 1 if c2 else 28):
        shape = (G, B, Ho, Wo) if G == 2 else (B, Ho, Wo)
        y = torch.full(shape + (cout + 8,), 7.0, dtype=dt, device=DEV)[..., :cout]
        ch = None
        if c2:
            y2 = torch.full(shape + (c2 + 8,), 7.0, dtype=dt, device=DEV)[..., :c2]
            ch = dict(w=w2p, kp=kp2, bias=b2p, y=y2, cout=c2, keep=False)
        run(ops.conv2d(xa, wp, kp, bp, y, 3, 3, 2, 2, 1, 1, cin, cout, ops.ACT_SILU, tile=tile, chain=ch))
        outs.append(ch["y"].clone() if c2 else y.clone())
    assert torch.equal(outs[0], outs[1]), f"cwide s2 != igemm, max diff {(outs[0].float() - outs[1].float()).abs().max().item()}"
    for g in range(G):
        m = F.silu(F.conv2d(q(xs[g], dt), q(ws[g], dt), bs[g], 2, 1))
        if c2:
            m = F.silu(F.conv2d(q(m, dt), q(w2s[g], dt), b2s[g]))
        close(from_act(outs[0][g] if G == 2 else outs[0]), m, dt, f"cwide s2 {case} group {g}", factor=2.0)


WREG_CASES = [
    # B, H, W, cin, cout, k, stride, act, use_res, groups, shape (tile id 60 + shape): weight operand fed from registers (igemm_wreg.hip)
    (4, 40, 40, 128, 128, 3, 1, ops.ACT_SILU, True, 2, 1),      # Bottleneck 3x3 + shortcut, both backbones: 18 K slices
    (3, 20, 20, 256, 256, 3, 1, ops.ACT_SILU, False, 1, 2),     # 128 x 256 tile, K = 36 slices
    (2, 40, 40, 128, 256, 3, 2, ops.ACT_SILU, False, 2, 2),     # stride 2
    (2, 23, 29, 64, 136, 3, 1, ops.ACT_GELU, True, 1, 1),       # ragged M and N (N tile 2 half empty), K = 9 slices (9 % 3 == 0)
    (1, 17, 19, 64, 192, 1, 1, ops.ACT_NONE, False, 1, 1),      # ONE K slice (tail path only), 1x1
    (2, 33, 31, 128, 384, 1, 1, ops.ACT_SILU, True, 1, 1),      # two K slices, three channel tiles
    (5, 20, 20, 512, 512, 1, 1, ops.ACT_SILU, False, 1, 2),     # 1x1, K = 8 slices (8 % 3 == 2), two 256-wide tiles
    (2, 20, 20, 1024, 512, 1, 1, ops.ACT_SILU, False, 2, 1),    # SPPF cv2, paired: K = 16 slices (16 % 3 == 1)
    (1, 13, 13, 64, 384, 3, 2, ops.ACT_SILU, False, 1, 2),      # 128 x 256 tile reaching beyond Cout: rejected (checked below)
    # round 4: a wave owns 64 channels (two weight buffers, the K loop unrolled by two): 128 x 512 with eight waves (shape 3), 128 x 256 with four (4)
    (3, 20, 20, 256, 512, 3, 1, ops.ACT_SILU, True, 1, 3),      # K = 36 slices (even), residual, ragged last pixel tile (1200 = 9 * 128 + 48)
    (2, 40, 40, 128, 512, 3, 2, ops.ACT_SILU, False, 2, 3),     # stride 2, paired, K = 18 slices
    (5, 20, 20, 512, 1024, 1, 1, ops.ACT_SILU, False, 1, 3),    # 1x1, two 512-wide tiles, K = 8 slices
    (2, 17, 19, 192, 512, 1, 1, ops.ACT_NONE, True, 1, 3),      # K = 3 slices (odd: the tail step), linear, residual
    (1, 9, 9, 64, 512, 1, 1, ops.ACT_SILU, False, 1, 3),        # ONE K slice, one ragged pixel tile
    (3, 20, 20, 256, 256, 3, 1, ops.ACT_SILU, True, 2, 4),      # four waves x 64 channels, paired, residual, K = 36 slices
    (2, 33, 31, 128, 512, 1, 1, ops.ACT_SILU, False, 1, 4),     # two 256-wide tiles, K = 2 slices
    (1, 13, 13, 320, 256, 3, 2, ops.ACT_NONE, False, 1, 4),     # K = 45 slices (odd), stride 2, linear
    # 64-pixel tiles (twice the workgroups for the layers with few pixels): 64 x 256 (shape 5), 64 x 128 (6)
    (3, 20, 20, 256, 256, 3, 1, ops.ACT_SILU, True, 2, 5),      # paired, residual, K = 36 slices, ragged last tile (1200 = 18 * 64 + 48)
    (2, 20, 20, 512, 512, 1, 1, ops.ACT_SILU, False, 1, 5),     # two 256-wide tiles, K = 8 slices
    (1, 11, 13, 192, 256, 3, 2, ops.ACT_NONE, False, 1, 5),     # K = 27 slices (odd), stride 2, linear
    (3, 20, 20, 256, 128, 3, 1, ops.ACT_SILU, True, 2, 6),      # 64 x 128: three weight buffers, K = 36 slices (36 % 3 == 0)
    (2, 20, 20, 512, 384, 1, 1, ops.ACT_SILU, False, 1, 6),     # three channel tiles, K = 8 slices (8 % 3 == 2)
    (1, 9, 9, 64, 136, 1, 1, ops.ACT_NONE, True, 1, 6),         # ragged M and N, ONE K slice
    (2, 10, 10, 256, 1024, 1, 1, ops.ACT_GELU, False, 2, 4),    # an MLP's fc1 (GELU epilogue) on the four-wave 64-channel tile, paired
    (2, 10, 10, 256, 512, 1, 1, ops.ACT_GELU, False, 1, 5),     # ... and on the 64-pixel tiles
    (1, 10, 10, 128, 384, 1, 1, ops.ACT_GELU, True, 1, 6),
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", WREG_CASES)
def test_conv_weights_from_registers_kernel(case, dt):
    """igemm_wreg.hip vs torch, and BIT-EXACT vs the implicit-GEMM kernel.  The K loop is unrolled by three with a tail of 0-2
    steps (by two with a tail of 0-1 for the 64-channel-per-wave tiles), the pixel ring is four stages deep, the weights come
    fragment-major straight from memory: the cases cover every tail length, 1x1 / 3x3 / stride 2, residuals, ragged tiles, all four
    tile shapes and the paired (groups = 2) launch."""
    B, H, W, cin, cout, k, st, act, use_res, G, shape = case
    p_ = k // 2
    Ho, Wo = (H + 2 * p_ - k) // st + 1, (W + 2 * p_ - k) // st + 1
    xs = [rnd((B, cin, H, W), 61 + g) for g in range(G)]
    ws = [rnd((cout, cin, k, k), 63 + g, 1.0 / math.sqrt(cin * k * k)) for g in range(G)]
    bs = [rnd((cout,), 65 + g, 0.2) for g in range(G)]
    rs = [rnd((B, cout, Ho, Wo), 67 + g) for g in range(G)] if use_res else None
    stk = (lambda t: torch.stack(t).contiguous()) if G == 2 else (lambda t: t[0])
    xa = stk([to_act(x, dt) for x in xs])
    packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
    wp, kp = stk([p0[0] for p0 in packs]), packs[0][1]
    bp = stk([ops.pack_bias(b.to(DEV), cout) for b in bs])
    ra = stk([to_act(r, dt) for r in rs]) if use_res else None
    ldy = -(-cout // 8) * 8 + 8
    outs = []
    for tile in (60 + shape, 2):
        ybuf = torch.full((G, B, Ho, Wo, ldy) if G == 2 else (B, Ho, Wo, ldy), 7.0, dtype=dt, device=DEV)
        y = ybuf[..., :cout]
        try:
            run(ops.conv2d(xa, wp, kp, bp, y, k, k, st, st, p_, p_, cin, cout, act, res=ra, alpha_acc=0.75, alpha_res=1.25, tile=tile))
        except ops._lib.IcafError as e:
            assert tile > 60 and "beyond the packed weights" in str(e) and cout == 384 and shape == 2, e
            return
        assert float((ybuf[..., cout:] - 7.0).abs().max()) == 0.0, "conv wrote outside its channel slice"
        outs.append(y.clone())
    assert not (cout == 384 and shape == 2 and cin == 64), "the 128 x 256 tile must reject Cout = 384 (Np = 384)"
    assert torch.equal(outs[0], outs[1]), f"wreg kernel != igemm, max diff {(outs[0].float() - outs[1].float()).abs().max().item()}"
    for g in range(G):
        ref = F.conv2d(q(xs[g], dt), q(ws[g], dt), bs[g], st, p_)
        ref = {ops.ACT_NONE: lambda t: t, ops.ACT_SILU: F.silu, ops.ACT_GELU: F.gelu}[act](ref) * 0.75
        if use_res:
            ref = ref + 1.25 * q(rs[g], dt)
        close(from_act(outs[0][g] if G == 2 else outs[0]), ref, dt, f"wreg {case} group {g}")


def test_frag_weights_layout():
    """ops.frag_weights: lane (hi * 32 + r) of block (nb, ks) holds w[nb * 32 + r][ks * 16 + hi * 8 : + 8] (icaf.h: icaf_conv_args.wf)."""
    w = torch.arange(128 * 64, dtype=torch.float32).reshape(128, 64).to(torch.bfloat16).to(DEV)
    f = ops.frag_weights(w)
    assert f.shape == (4, 4, 64, 8)
    for nb, ks, lane in ((0, 0, 0), (1, 2, 37), (3, 3, 63), (2, 1, 31)):
        hi, r = lane // 32, lane % 32
        assert torch.equal(f[nb, ks, lane], w[nb * 32 + r, ks * 16 + hi * 8: ks * 16 + hi * 8 + 8])
    w2 = torch.stack((w, w + 1)).contiguous()
    f2 = ops.frag_weights(w2)
    assert f2.shape == (2, 4, 4, 64, 8) and torch.equal(f2[1, 1, 2, 37], w2[1, 32 + 5, 32 + 8: 32 + 16])


CTILE_CASES = [
    # B, H, W, cin, cout, stride, use_res, ctile shape (tile id 40 + shape)
    (2, 40, 70, 16, 32, 1, False, 1),     # stem-like: 16 channels = 32 bytes per pixel, one tap per MFMA step
    (1, 37, 45, 32, 32, 1, True, 1),      # ragged patch grid, residual (Bottleneck cv2)
    (2, 24, 64, 32, 64, 1, False, 2),
    (1, 19, 33, 64, 64, 1, True, 2),
    (2, 40, 40, 64, 64, 1, True, 3),      # 8x16 patches: sub-tile spans two rows
    (1, 21, 50, 32, 48, 1, False, 3),     # Cout < BN
    (2, 48, 80, 32, 64, 2, False, 4),     # stride 2: even / odd column planes
    (1, 37, 41, 16, 64, 2, False, 4),     # odd input size, stride 2
    (1, 24, 24, 128, 128, 1, True, 5),
]


# (a pixel wider than 256 bytes stays on igemm: the fp32 form of the 128-channel case is not generated)
CTILE_PARAMS = [pytest.param(c, dt, id=f"case{i}-{str(dt).split('.')[-1]}") for i, c in enumerate(CTILE_CASES) for dt in DTYPES
                if c[3] * (4 if dt == torch.float32 else 2) <= 256]


@pytest.mark.parametrize("case,dt", CTILE_PARAMS)
def test_conv3x3_halo_tile_kernel(case, dt):
    """ctile.hip (3x3 direct convolution from an LDS halo patch) vs torch, and BIT-EXACT vs the implicit-GEMM kernel:
    both walk K in the same order with the same MFMA step, so any difference is an addressing bug."""
    B, H, W, cin, cout, s, use_res, shape = case
    x = rnd((B, cin, H, W), 11)
    w = rnd((cout, cin, 3, 3), 12, 1.0 / math.sqrt(cin * 9))
    bias = rnd((cout,), 13, 0.2)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    res = rnd((B, cout, Ho, Wo), 14) if use_res else None
    xa = to_act(x, dt, pad_to=cin + 16)
    wp, kp = ops.pack_conv_weight(w.to(DEV), dt)
    bp = ops.pack_bias(bias.to(DEV), cout)
    ra = to_act(res, dt) if use_res else None
    outs = []
    for tile in (40 + shape, 2 if cout > 32 else 3):
        ybuf = torch.zeros((B, Ho, Wo, cout + 8), dtype=dt, device=DEV)
        y = ybuf[..., :cout]
        run(ops.conv2d(xa, wp, kp, bp, y, 3, 3, s, s, 1, 1, cin, cout, ops.ACT_SILU, res=ra, alpha_acc=0.75,
                       alpha_res=1.25, tile=tile))
        assert float(ybuf[..., cout:].abs().max()) == 0.0
        outs.append(y.clone())
    ref = F.silu(F.conv2d(q(x, dt), q(w, dt), bias, s, 1)) * 0.75
    if use_res:
        ref = ref + 1.25 * q(res, dt)
    close(from_act(outs[0]), ref, dt, f"ctile {case}")
    assert torch.equal(outs[0], outs[1]), "halo-tile kernel and implicit GEMM must agree bit for bit"


BNECK_CASES = [
    # B, H, W, c, add, patch shape, paired streams
    (2, 40, 70, 32, True, 1, False),
    (1, 37, 45, 32, False, 1, True),
    (2, 24, 64, 64, True, 2, True),
    (1, 19, 33, 64, True, 3, False),
    (2, 40, 40, 64, False, 3, True),
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", BNECK_CASES)
def test_fused_bottleneck_equals_two_launches(case, dt):
    """icaf_bottleneck (1x1 on the LDS halo patch -> 3x3 -> + x, one launch) vs the same two layers as separate
    implicit-GEMM launches: bit-identical (same K order, same rounding points), and close to fp32 torch."""
    B, H, W, c, add, shape, paired = case
    G = 2 if paired else 1
    xs = [rnd((B, c, H, W), 41 + g) for g in range(G)]
    w1 = [rnd((c, c, 1, 1), 43 + g, 1.0 / math.sqrt(c)) for g in range(G)]
    w2 = [rnd((c, c, 3, 3), 45 + g, 1.0 / math.sqrt(9 * c)) for g in range(G)]
    b1 = [rnd((c,), 47 + g, 0.2) for g in range(G)]
    b2 = [rnd((c,), 49 + g, 0.2) for g in range(G)]

    def stack(ts):
        return torch.stack(ts).contiguous() if paired else ts[0]
    xa = stack([to_act(x, dt) for x in xs])
    p1 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w1]
    p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2]
    w1p, w2p = stack([p[0] for p in p1]), stack([p[0] for p in p2])
    b1p, b2p = stack([ops.pack_bias(b.to(DEV), c) for b in b1]), stack([ops.pack_bias(b.to(DEV), c) for b in b2])
    y_f, y_u, t = torch.zeros_like(xa), torch.zeros_like(xa), torch.zeros_like(xa)
    run(ops.bottleneck(xa, w1p, p1[0][1], b1p, w2p, p2[0][1], b2p, y_f, c, add, shape))
    run(ops.conv2d(xa, w1p, p1[0][1], b1p, t, 1, 1, 1, 1, 0, 0, c, c, ops.ACT_SILU))
    run(ops.conv2d(t, w2p, p2[0][1], b2p, y_u, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, res=xa if add else None))
    assert torch.equal(y_f, y_u), "fused Bottleneck must match the two-launch form bit for bit"
    for g in range(G):
        tt = q(F.silu(F.conv2d(q(xs[g], dt), q(w1[g], dt), b1[g])), dt)
        ref = F.silu(F.conv2d(tt, q(w2[g], dt), b2[g], 1, 1)) + (q(xs[g], dt) if add else 0)
        close(from_act(y_f[g] if paired else y_f), ref, dt, f"bottleneck {case} stream {g}")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 64, 96, 64, True, True), (1, 70, 90, 64, True, False), (3, 72, 64, 64, False, True),
                                  (1, 160, 160, 64, False, False)])
def test_bottleneck_with_chained_cv3_equals_two_launches(case, dt):
    """icaf_bottleneck with the C3's cv3 riding on it (the Bottleneck output and cat(m, cv2) stay in LDS) vs icaf_bottleneck
    followed by the 1x1 cv3 over the [cv2 | m] slots of the C3 buffer: bit-identical; ragged patches, shortcut on / off,
    one and two streams."""
    B, H, W, c3, add, paired = case
    c = 32
    G = 2 if paired else 1
    xs = [rnd((B, 3 * c, H, W), 201 + g) for g in range(G)]            # slots [a | b | a'] of the C3 buffer (a' overwritten)
    w1 = [rnd((c, c, 1, 1), 203 + g, 1.0 / math.sqrt(c)) for g in range(G)]
    w2 = [rnd((c, c, 3, 3), 205 + g, 1.0 / math.sqrt(9 * c)) for g in range(G)]
    w3 = [rnd((c3, 2 * c, 1, 1), 207 + g, 1.0 / math.sqrt(2 * c)) for g in range(G)]      # columns [m | cv2], as C3.cv3's
    b1, b2 = [rnd((c,), 209 + g, 0.2) for g in range(G)], [rnd((c,), 211 + g, 0.2) for g in range(G)]
    b3 = [rnd((c3,), 213 + g, 0.2) for g in range(G)]
    st = (lambda ts: torch.stack(ts).contiguous()) if paired else (lambda ts: ts[0])
    cat = st([to_act(x, dt) for x in xs])
    p1 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w1]
    p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2]
    p3 = [ops.pack_conv_weight(torch.cat((w[:, c:], w[:, :c]), 1).to(DEV), dt) for w in w3]     # -> [cv2 | m]
    w1p, w2p, w3p = (st([p_[0] for p_ in ps]) for ps in (p1, p2, p3))
    b1p, b2p, b3p = (st([ops.pack_bias(b.to(DEV), n) for b in bs]) for bs, n in ((b1, c), (b2, c), (b3, c3)))
    a, bhalf, a2 = cat[..., :c], cat[..., c:2 * c], cat[..., 2 * c:]
    lead = (G, B) if paired else (B,)
    y_u = torch.zeros((*lead, H, W, c3), dtype=dt, device=DEV)
    y_f = torch.full((*lead, H, W, c3 + 8), 7.0, dtype=dt, device=DEV)
    run(ops.bottleneck(a, w1p, p1[0][1], b1p, w2p, p2[0][1], b2p, a2, c, add, 1))
    run(ops.conv2d(cat[..., c:], w3p, p3[0][1], b3p, y_u, 1, 1, 1, 1, 0, 0, 2 * c, c3, ops.ACT_SILU))
    m_out = a2.clone()
    a2.fill_(3.0)                                                      # the fused launch must not need (or write) slot a'
    run(ops.bottleneck(a, w1p, p1[0][1], b1p, w2p, p2[0][1], b2p, None, c, add, 1,
                       cv3=dict(w=w3p, kp=p3[0][1], bias=b3p, y=y_f[..., :c3], cout=c3, x2=bhalf)))
    assert torch.equal(y_f[..., :c3], y_u)
    assert bool((y_f[..., c3:] == 7.0).all()) and bool((a2 == 3.0).all())
    for g in range(G):
        x = q(xs[g], dt)
        t = q(F.silu(F.conv2d(x[:, :c], q(w1[g], dt), b1[g])), dt)
        mref = q(F.silu(F.conv2d(t, q(w2[g], dt), b2[g], 1, 1)) + (x[:, :c] if add else 0), dt)
        ref = F.silu(F.conv2d(torch.cat((mref, x[:, c:2 * c]), 1), q(w3[g], dt), b3[g]))
        close(from_act(y_f[g][..., :c3] if paired else y_f[..., :c3]), ref, dt, f"bneck+cv3 {case} stream {g}", factor=2.0)
        close(from_act(m_out[g] if paired else m_out), mref, dt, f"bneck {case} stream {g}", factor=2.0)


CHAIN_CASES = [
    # B, H, W, cin, c1 (3x3 layer out), c2 (chained 1x1 out), stride, tile, paired
    (2, 40, 48, 32, 64, 64, 2, 0, False),      # backbone row 1 -> C3 cv1|cv2 (yolov5s)
    (1, 33, 37, 64, 64, 48, 1, 22, True),      # ragged, narrower second layer
    (2, 24, 24, 64, 128, 128, 2, 0, True),     # row 3 -> C3
    (1, 20, 28, 128, 128, 96, 1, 21, False),
    (1, 19, 23, 64, 96, 128, 1, 1, False),     # first layer narrower than the tile
]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CHAIN_CASES)
def test_conv_with_chained_1x1_equals_two_launches(case, dt):
    """3x3 conv + SiLU with a chained 1x1 conv + SiLU on the LDS-resident output tile (one launch, the intermediate tensor
    never written) vs the two layers as separate launches: bit-identical, and close to fp32 torch."""
    B, H, W, cin, c1, c2, st, tile, paired = case
    G = 2 if paired else 1
    xs = [rnd((B, cin, H, W), 81 + g) for g in range(G)]
    w1 = [rnd((c1, cin, 3, 3), 83 + g, 1.0 / math.sqrt(9 * cin)) for g in range(G)]
    w2 = [rnd((c2, c1, 1, 1), 85 + g, 1.0 / math.sqrt(c1)) for g in range(G)]
    b1 = [rnd((c1,), 87 + g, 0.2) for g in range(G)]
    b2 = [rnd((c2,), 89 + g, 0.2) for g in range(G)]
    stk = (lambda ts: torch.stack(ts).contiguous()) if paired else (lambda ts: ts[0])
    xa = stk([to_act(x, dt) for x in xs])
    p1 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w1]
    p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2]
    w1p, w2p = stk([p_[0] for p_ in p1]), stk([p_[0] for p_ in p2])
    b1p, b2p = stk([ops.pack_bias(b.to(DEV), c1) for b in b1]), stk([ops.pack_bias(b.to(DEV), c2) for b in b2])
    Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
    shp = (G, B, Ho, Wo) if paired else (B, Ho, Wo)
    mid = torch.zeros((*shp, c1), dtype=dt, device=DEV)
    y_u = torch.zeros((*shp, c2 + 8), dtype=dt, device=DEV)[..., :c2]
    y_f = torch.zeros((*shp, c2 + 8), dtype=dt, device=DEV)[..., :c2]
    run(ops.conv2d(xa, w1p, p1[0][1], b1p, mid, 3, 3, st, st, 1, 1, cin, c1, ops.ACT_SILU))
    run(ops.conv2d(mid, w2p, p2[0][1], b2p, y_u, 1, 1, 1, 1, 0, 0, c1, c2, ops.ACT_SILU))
    dummy = torch.zeros((*shp, c1), dtype=dt, device=DEV)
    run(ops.conv2d(xa, w1p, p1[0][1], b1p, dummy, 3, 3, st, st, 1, 1, cin, c1, ops.ACT_SILU, tile=tile,
                   chain=dict(w=w2p, kp=p2[0][1], bias=b2p, y=y_f, cout=c2)))
    assert float(dummy.abs().max()) == 0.0, "the intermediate tensor must not be written"
    assert torch.equal(y_f, y_u)
    y_k = torch.zeros_like(y_f)
    run(ops.conv2d(xa, w1p, p1[0][1], b1p, dummy, 3, 3, st, st, 1, 1, cin, c1, ops.ACT_SILU, tile=tile,
                   chain=dict(w=w2p, kp=p2[0][1], bias=b2p, y=y_k, cout=c2, keep=True)))
    assert torch.equal(dummy, mid) and torch.equal(y_k, y_u), "chain_keep writes both layers' outputs"
    for g in range(G):
        t = q(F.silu(F.conv2d(q(xs[g], dt), q(w1[g], dt), b1[g], st, 1)), dt)
        ref = F.silu(F.conv2d(t, q(w2[g], dt), b2[g]))
        close(from_act(y_f[g] if paired else y_f), ref, dt, f"chain {case} stream {g}")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 40, 48, 64, 0, True), (1, 33, 37, 64, 22, False), (2, 20, 24, 128, 0, True), (1, 21, 19, 128, 21, False)])
def test_bottleneck_3x3_with_shortcut_chained_to_next_1x1(case, dt):
    """A Bottleneck's 3x3 conv + shortcut (written IN PLACE over its own residual) with the next Bottleneck's 1x1 chained on the
    tile (chain_keep + res): both outputs bit-identical to the two launches."""
    B, H, W, c, tile, paired = case
    G = 2 if paired else 1
    ts = [rnd((B, c, H, W), 231 + g) for g in range(G)]                # the 1x1's output feeding the 3x3
    rs = [rnd((B, c, H, W), 233 + g) for g in range(G)]                # block input = residual, overwritten by the output
    w1 = [rnd((c, c, 3, 3), 235 + g, 1.0 / math.sqrt(9 * c)) for g in range(G)]
    w2 = [rnd((c, c, 1, 1), 237 + g, 1.0 / math.sqrt(c)) for g in range(G)]
    b1, b2 = [rnd((c,), 239 + g, 0.2) for g in range(G)], [rnd((c,), 241 + g, 0.2) for g in range(G)]
    stk = (lambda xs: torch.stack(xs).contiguous()) if paired else (lambda xs: xs[0])
    ta = stk([to_act(t, dt) for t in ts])
    p1 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w1]
    p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2]
    w1p, w2p = stk([p_[0] for p_ in p1]), stk([p_[0] for p_ in p2])
    b1p, b2p = stk([ops.pack_bias(b.to(DEV), c) for b in b1]), stk([ops.pack_bias(b.to(DEV), c) for b in b2])
    a_u, a_f = stk([to_act(r, dt) for r in rs]), stk([to_act(r, dt) for r in rs])
    n_u, n_f = torch.zeros_like(a_u), torch.zeros_like(a_u)
    run(ops.conv2d(ta, w1p, p1[0][1], b1p, a_u, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, res=a_u))
    run(ops.conv2d(a_u, w2p, p2[0][1], b2p, n_u, 1, 1, 1, 1, 0, 0, c, c, ops.ACT_SILU))
    run(ops.conv2d(ta, w1p, p1[0][1], b1p, a_f, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, res=a_f, tile=tile,
                   chain=dict(w=w2p, kp=p2[0][1], bias=b2p, y=n_f, cout=c, keep=True)))
    assert torch.equal(a_f, a_u) and torch.equal(n_f, n_u)
    for g in range(G):
        m = q(F.silu(F.conv2d(q(ts[g], dt), q(w1[g], dt), b1[g], 1, 1)), dt) + q(rs[g], dt)
        close(from_act(a_f[g] if paired else a_f), m, dt, f"3x3+res {case}", factor=2.0)
        close(from_act(n_f[g] if paired else n_f), F.silu(F.conv2d(q(m, dt), q(w2[g], dt), b2[g])), dt, f"next 1x1 {case}", factor=2.0)


def test_conv3x3_halo_tile_rejects_other_layers():
    x = torch.zeros((1, 8, 8, 32), dtype=torch.bfloat16, device=DEV)
    w = rnd((32, 32, 1, 1), 1)
    wp, kp = ops.pack_conv_weight(w.to(DEV), torch.bfloat16)
    y = torch.zeros((1, 8, 8, 32), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ops._lib.IcafError):
        run(ops.conv2d(x, wp, kp, None, y, 1, 1, 1, 1, 0, 0, 32, 32, ops.ACT_SILU, tile=41))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", [(2, 11, 15, 2, 200, 0), (1, 7, 9, 3, 64, 2), (2, 10, 10, 2, 40, 22)])
def test_conv2d_nearest_pre_term_is_conv_over_upsampled_concat(case, dt):
    """SiLU(W . cat(up(a), b) + bias) computed as SiLU(up(Wa . a) + Wb . b + bias): the a-half at LOW resolution (fp32 out)
    enters the GEMM over b as its nearest-resized pre-activation term (pre_mode 1) — the head's Upsample -> Concat -> C3."""
    B, h, w, s, cout, tile = case
    ca, cb, H, W = 64, 128, h * s, w * s
    a_, b_ = rnd((B, ca, h, w), 41), rnd((B, cb, H, W), 42)
    wt = rnd((cout, ca + cb, 1, 1), 43, 1.0 / math.sqrt(ca + cb))
    bias = rnd((cout,), 44, 0.2)
    aa, ba = to_act(a_, dt), to_act(b_, dt, pad_to=cb + 32)
    wa, kpa = ops.pack_conv_weight(wt[:, :ca].to(DEV), dt)
    wb, kpb = ops.pack_conv_weight(wt[:, ca:].to(DEV), dt)
    bp = ops.pack_bias(bias.to(DEV), cout)
    P = torch.zeros((B, h, w, cout), dtype=torch.float32, device=DEV)
    y = torch.zeros((B, H, W, cout), dtype=dt, device=DEV)
    run(ops.conv2d(aa, wa, kpa, None, P, 1, 1, 1, 1, 0, 0, ca, cout, ops.ACT_NONE))
    run(ops.conv2d(ba, wb, kpb, bp, y, 1, 1, 1, 1, 0, 0, cb, cout, ops.ACT_SILU, pre=P, pre_nearest=True, tile=tile))
    cat = torch.cat((F.interpolate(q(a_, dt), scale_factor=s, mode="nearest"), q(b_, dt)), 1)
    close(from_act(y), F.silu(F.conv2d(cat, q(wt, dt), bias)), dt, f"nearest pre term {case}")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(3, 80, 80, 256, 128, 20, 20, False), (2, 37, 45, 128, 200, 7, 9, False), (2, 40, 40, 512, 256, 16, 16, False),
                                  (1, 20, 20, 1024, 512, 10, 10, False), (2, 22, 30, 128, 256, 11, 15, True)])
def test_streaming_kernel_pre_term_is_bit_identical_to_igemm(case, dt):
    """The pre-activation term (bilinear / nearest) as an epilogue policy of the persistent streaming GEMM: same bits as igemm."""
    B, H, W, cin, cout, th, tw, nearest = case
    x = rnd((B, cin, H, W), 131)
    w = rnd((cout, cin, 1, 1), 132, 1.0 / math.sqrt(cin))
    bias = rnd((cout,), 133, 0.2)
    pre = rnd((B, cout, th, tw), 134)
    xa = to_act(x, dt)
    wp, kp = ops.pack_conv_weight(w.to(DEV), dt)
    bp = ops.pack_bias(bias.to(DEV), cout)
    pa = pre.permute(0, 2, 3, 1).contiguous().to(DEV)
    outs = []
    for tile in (51, 52, 1, 22):
        y = torch.zeros((B, H, W, cout), dtype=dt, device=DEV)
        run(ops.conv2d(xa, wp, kp, bp, y, 1, 1, 1, 1, 0, 0, cin, cout, ops.ACT_SILU, pre=pa, pre_nearest=nearest, tile=tile))
        outs.append(y)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    mode = dict(mode="nearest") if nearest else dict(mode="bilinear", align_corners=False)
    ref = F.silu(F.conv2d(q(x, dt), q(w, dt), bias) + F.interpolate(pre, size=(H, W), **mode))
    close(from_act(outs[0]), ref, dt, f"streaming kernel + pre term {case}")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("tile", [0, 2, 22])
def test_conv2d_pre_activation_bilinear_term(dt, tile):
    """y = act(conv(x) + bias + F.interpolate(pre, bilinear, align_corners=False)): the epilogue term behind DMFF's
    fused tail (reference models/common.py:827-841)."""
    B, H, W, cin, cout, th, tw = 2, 22, 30, 64, 40, 5, 7
    x = rnd((B, cin, H, W), 31)
    w = rnd((cout, cin, 1, 1), 32, 1.0 / math.sqrt(cin))
    bias = rnd((cout,), 33, 0.2)
    pre = rnd((B, cout, th, tw), 34)
    xa = to_act(x, dt)
    wp, kp = ops.pack_conv_weight(w.to(DEV), dt)
    bp = ops.pack_bias(bias.to(DEV), cout)
    pa = pre.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.zeros((B, H, W, cout), dtype=dt, device=DEV)
    run(ops.conv2d(xa, wp, kp, bp, y, 1, 1, 1, 1, 0, 0, cin, cout, ops.ACT_SILU, pre=pa, tile=tile))
    ref = F.silu(F.conv2d(q(x, dt), q(w, dt), bias) + F.interpolate(pre, size=(H, W), mode="bilinear", align_corners=False))
    close(from_act(y), ref, dt, "conv + bilinear pre-activation term")


@pytest.mark.parametrize("dt", DTYPES)
def test_conv2d_fp32_output_and_groups(dt):
    rows, cin, cout = 300, 64, 192
    x = rnd((2, rows, cin), 5)
    w = rnd((2, cout, cin), 6, 1.0 / math.sqrt(cin))
    b = rnd((2, cout), 7, 0.1)
    res = rnd((2, rows, cout), 8)
    xg = x.to(DEV).to(dt).contiguous()
    rg = res.to(DEV).to(dt).contiguous()
    packs = [ops.pack_matrix(w[g].to(DEV), dt) for g in range(2)]
    wp = torch.stack([p_[0] for p_ in packs]).contiguous()
    bp = torch.stack([ops.pack_bias(b[g].to(DEV), cout) for g in range(2)]).contiguous()
    y = torch.zeros((2, rows, cout), dtype=dt, device=DEV)
    gs = dict(x=xg.stride(0), w=wp.stride(0), bias=bp.stride(0), y=y.stride(0), res=rg.stride(0))
    run(ops.conv2d(xg[0].view(rows, 1, 1, cin), wp, packs[0][1], bp, y[0].view(rows, 1, 1, cout), 1, 1, 1, 1, 0, 0, cin,
                   cout, ops.ACT_GELU, res=rg[0].view(rows, 1, 1, cout), alpha_acc=(0.5, 2.0), alpha_res=(1.5, -1.0),
                   groups=2, group_strides=gs))
    for g, (aa, ar) in enumerate(((0.5, 1.5), (2.0, -1.0))):
        ref = aa * F.gelu(q(x[g], dt) @ q(w[g], dt).t() + b[g]) + ar * q(res[g], dt)
        close(y[g].float().cpu(), ref, dt, f"grouped linear g={g}")
    # fp32 output from a 16-bit compute type (Detect convs)
    y32 = torch.zeros((rows, 1, 1, 20), dtype=torch.float32, device=DEV)
    w2 = rnd((18, cin), 9, 0.2)
    wp2, kp2 = ops.pack_matrix(w2.to(DEV), dt)
    run(ops.conv2d(xg[0].view(rows, 1, 1, cin), wp2, kp2, None, y32[..., :18], 1, 1, 1, 1, 0, 0, cin, 18, ops.ACT_NONE))
    close(y32[:, 0, 0, :18].cpu(), q(x[0], dt) @ q(w2, dt).t(), dt, "fp32-out linear")


@pytest.mark.parametrize("dt", DTYPES)
def test_first_layer_space_to_depth(dt):
    """6x6/s2/p2 conv over the image == preprocess(mode 1) + 3x3/s1/p1 conv with re-indexed weights."""
    B, H, W, cout = 2, 64, 96, 32
    img = torch.from_numpy(np.random.default_rng(1).random((B, 3, H, W), dtype=np.float32))
    w = rnd((cout, 3, 6, 6), 2, 0.1)
    bias = rnd((cout,), 3, 0.1)
    vec = ops.VEC[dt]
    cpad = -(-12 // vec) * vec
    pre = torch.zeros((B, H // 2, W // 2, cpad), dtype=dt, device=DEV)
    run(ops.preprocess(img.to(DEV), pre, 1))
    wp, kp = ops.pack_conv_weight(ops.s2d_conv_weight(w.to(DEV)), dt, cpad)
    y = torch.zeros((B, H // 2, W // 2, cout), dtype=dt, device=DEV)
    run(ops.conv2d(pre, wp, kp, ops.pack_bias(bias.to(DEV), cout), y, 3, 3, 1, 1, 1, 1, cpad, cout, ops.ACT_SILU))
    ref = F.silu(F.conv2d(q(img, dt), q(w, dt), bias, 2, 2))
    close(from_act(y), ref, dt, "stem conv")
    # mode 0 (plain pad)
    pad = torch.zeros((B, H, W, vec), dtype=dt, device=DEV)
    run(ops.preprocess(img.to(DEV), pad, 0))
    assert torch.equal(pad[..., :3].float().cpu(), q(img, dt).permute(0, 2, 3, 1))
    assert float(pad[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mode", [0, 1])
def test_preprocess_u8_matches_float_path(dt, mode):
    """uint8 6-channel staging == reference recipe (`.float() / 255`, split, then the fp32 staging kernel), bit for bit."""
    B, H, W = 2, 12, 20
    g = np.random.default_rng(7)
    img6 = torch.from_numpy(g.integers(0, 256, (B, 6, H, W), dtype=np.uint8)).to(DEV)
    cpad = 16 if mode == 1 else 8
    Ho, Wo = (H // 2, W // 2) if mode == 1 else (H, W)
    pair = torch.zeros((2, B, Ho, Wo, cpad), dtype=dt, device=DEV)
    run(ops.preprocess_u8(img6, pair, mode))
    f = (img6.cpu().float() / 255.0).to(DEV)      # true division on the CPU, as the reference's CPU path (torch's GPU
    for s_, sl in enumerate((slice(0, 3), slice(3, 6))):     # scalar division multiplies by the reciprocal instead)
        ref = torch.zeros((B, Ho, Wo, cpad), dtype=dt, device=DEV)
        run(ops.preprocess(f[:, sl].contiguous(), ref, mode))
        assert torch.equal(pair[s_], ref)
    single = torch.zeros((B, Ho, Wo, cpad), dtype=dt, device=DEV)
    run(ops.preprocess_u8(img6, single, mode, c0=3))
    assert torch.equal(single, pair[1])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 48, 96, 32, True, False), (1, 44, 72, 64, False, False), (2, 64, 128, 32, True, True),
                                  (3, 36, 40, 64, True, True), (2, 96, 200, 64, True, False), (4, 160, 160, 64, True, True)])      # (64 channels: the tile is written back in two halves)
def test_persistent_stem_equals_staging_plus_conv(case, dt):
    """icaf_stem (staging + 6x6/s2 conv in one persistent kernel, image read directly) vs icaf_preprocess_* followed by
    the 3x3 space-to-depth convolution on the implicit-GEMM kernel: bit-identical."""
    B, H, W, cout, paired, u8 = case
    G = 2 if paired else 1
    g = np.random.default_rng(61)
    ws = [rnd((cout, 3, 6, 6), 62 + i, 1.0 / math.sqrt(108)) for i in range(G)]
    bs = [rnd((cout,), 64 + i, 0.2) for i in range(G)]
    packs = [ops.pack_conv_weight(ops.s2d_conv_weight(w.to(DEV)), dt, 16) for w in ws]
    st = (lambda ts: torch.stack(ts).contiguous()) if paired else (lambda ts: ts[0])
    wp, kp, bp = st([p_[0] for p_ in packs]), packs[0][1], st([ops.pack_bias(b.to(DEV), cout) for b in bs])
    if u8:
        img = torch.from_numpy(g.integers(0, 256, (B, 6, H, W), dtype=np.uint8)).to(DEV)
    else:
        img = torch.from_numpy(g.random((G, B, 3, H, W), dtype=np.float32)).to(DEV)
        if not paired:
            img = img[0].contiguous()
    shape = (G, B, H // 2, W // 2) if paired else (B, H // 2, W // 2)
    y_f = torch.zeros((*shape, cout), dtype=dt, device=DEV)
    y_u = torch.zeros_like(y_f)
    pre = torch.zeros((*shape, 16), dtype=dt, device=DEV)
    run(ops.stem(img, wp, kp, bp, y_f, cout))
    run(ops.preprocess_u8(img, pre, 1) if u8 else ops.preprocess(img, pre, 1))
    run(ops.conv2d(pre, wp, kp, bp, y_u, 3, 3, 1, 1, 1, 1, 16, cout, ops.ACT_SILU))
    assert torch.equal(y_f, y_u)
    x0 = (img[:, :3].cpu().float() / 255.0) if u8 else (img[0] if paired else img).cpu()
    ref = F.silu(F.conv2d(q(x0, dt), q(ws[0], dt), bs[0], 2, 2))
    close(from_act(y_f[0] if paired else y_f), ref, dt, f"stem {case}")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 64, 256, 64, True, False), (1, 44, 72, 48, False, False), (2, 136, 130, 64, True, True),
                                  (3, 36, 40, 64, True, True), (1, 320, 320, 64, True, False), (2, 52, 130, 64, False, False), (2, 100, 644, 64, True, False)])
def test_stem2_equals_stem_then_chained_conv(case, dt):
    """icaf_stem2 (stem + 3x3/s2 conv + chained 1x1 in one persistent kernel; the two intermediate tensors stay in LDS) vs
    icaf_stem followed by icaf_conv2d with the chained 1x1: bit-identical, and close to fp32 torch.  Sizes cover tiles cut
    by the image border, odd stem / output sizes (H/2 or W/2 odd), one and two streams, uint8 and fp32 images."""
    B, H, W, c2, paired, u8 = case
    G = 2 if paired else 1
    g = np.random.default_rng(161)
    w0 = [rnd((32, 3, 6, 6), 162 + i, 1.0 / math.sqrt(108)) for i in range(G)]
    w1 = [rnd((64, 32, 3, 3), 164 + i, 1.0 / math.sqrt(288)) for i in range(G)]
    w2 = [rnd((c2, 64, 1, 1), 166 + i, 1.0 / 8) for i in range(G)]
    b0 = [rnd((32,), 168 + i, 0.2) for i in range(G)]
    b1 = [rnd((64,), 170 + i, 0.2) for i in range(G)]
    b2 = [rnd((c2,), 172 + i, 0.2) for i in range(G)]
    st = (lambda ts: torch.stack(ts).contiguous()) if paired else (lambda ts: ts[0])
    p0 = [ops.pack_conv_weight(ops.s2d_conv_weight(w.to(DEV)), dt, 16) for w in w0]
    p1 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w1]
    p2 = [ops.pack_conv_weight(w.to(DEV), dt) for w in w2]
    w0p, w1p, w2p = (st([p_[0] for p_ in ps]) for ps in (p0, p1, p2))
    b0p, b1p, b2p = (st([ops.pack_bias(b.to(DEV), n) for b in bs]) for bs, n in ((b0, 32), (b1, 64), (b2, c2)))
    if u8:
        img = torch.from_numpy(g.integers(0, 256, (B, 6, H, W), dtype=np.uint8)).to(DEV)
    else:
        img = torch.from_numpy(g.random((G, B, 3, H, W), dtype=np.float32)).to(DEV)
        if not paired:
            img = img[0].contiguous()
    Hs, Ws = H // 2, W // 2
    Ho, Wo = (Hs - 1) // 2 + 1, (Ws - 1) // 2 + 1
    lead = (G, B) if paired else (B,)
    t0 = torch.zeros((*lead, Hs, Ws, 32), dtype=dt, device=DEV)
    dummy = torch.zeros((*lead, Ho, Wo, 64), dtype=dt, device=DEV)
    y_u = torch.zeros((*lead, Ho, Wo, c2 + 8), dtype=dt, device=DEV)[..., :c2]
    y_f = torch.full((*lead, Ho, Wo, c2 + 8), 7.0, dtype=dt, device=DEV)
    run(ops.stem(img, w0p, p0[0][1], b0p, t0, 32))
    run(ops.conv2d(t0, w1p, p1[0][1], b1p, dummy, 3, 3, 2, 2, 1, 1, 32, 64, ops.ACT_SILU,
                   chain=dict(w=w2p, kp=p2[0][1], bias=b2p, y=y_u, cout=c2)))
    run(ops.stem2(img, w0p, p0[0][1], b0p, w1p, p1[0][1], b1p, w2p, p2[0][1], b2p, y_f[..., :c2], 32, 64, c2))
    assert torch.equal(y_f[..., :c2], y_u)
    assert bool((y_f[..., c2:] == 7.0).all()), "channels beyond C2 must not be written"
    x0 = (img[:, :3].cpu().float() / 255.0) if u8 else (img[0] if paired else img).cpu()
    t = q(F.silu(F.conv2d(q(x0, dt), q(w0[0], dt), b0[0], 2, 2)), dt)
    t = q(F.silu(F.conv2d(t, q(w1[0], dt), b1[0], 2, 1)), dt)
    ref = F.silu(F.conv2d(t, q(w2[0], dt), b2[0]))
    close(from_act(y_f[0][..., :c2] if paired else y_f[..., :c2]), ref, dt, f"stem2 {case}", factor=2.0)


def test_stem2_rejects_other_widths():
    img = torch.zeros((1, 3, 32, 32), dtype=torch.float32, device=DEV)
    dt = torch.bfloat16
    w0, kp0 = ops.pack_conv_weight(ops.s2d_conv_weight(torch.zeros((32, 3, 6, 6), device=DEV)), dt, 16)
    w1, kp1 = ops.pack_conv_weight(torch.zeros((64, 32, 3, 3), device=DEV), dt)
    w2, kp2 = ops.pack_conv_weight(torch.zeros((128, 64, 1, 1), device=DEV), dt)
    b = ops.pack_bias(torch.zeros(128, device=DEV), 128)
    y = torch.zeros((1, 8, 8, 128), dtype=dt, device=DEV)
    with pytest.raises(ops._lib.IcafError, match="built for 32 -> 64"):
        run(ops.stem2(img, w0, kp0, b, w1, kp1, b, w2, kp2, b, y, 32, 64, 128))


@pytest.mark.parametrize("dt", DTYPES)
def test_sppf_upsample_copy(dt):
    x = rnd((2, 64, 20, 12), 11)
    cat = torch.zeros((2, 20, 12, 256), dtype=dt, device=DEV)
    cat[..., :64] = to_act(x, dt)
    run(ops.sppf_pool(cat[..., :64], cat[..., 64:128], cat[..., 128:192], cat[..., 192:], 5))
    y1 = F.max_pool2d(q(x, dt), 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    assert torch.equal(from_act(cat), torch.cat((q(x, dt), y1, y2, y3), 1))
    up = torch.zeros((2, 40, 24, 96), dtype=dt, device=DEV)
    run(ops.upsample_nearest(cat[..., :64], up[..., 32:], 2))
    assert torch.equal(from_act(up[..., 32:]), F.interpolate(q(x, dt), scale_factor=2, mode="nearest"))
    assert float(up[..., :32].abs().max()) == 0.0
    cp = torch.zeros((2, 20, 12, 80), dtype=dt, device=DEV)
    run(ops.copy_channels(cat[..., 64:128], cp[..., 16:]))
    assert torch.equal(from_act(cp[..., 16:]), y1)


POOL_CASES = [(2, 128, 40, 40, 20, 20), (1, 64, 40, 40, 16, 16), (1, 64, 64, 80, 20, 20), (2, 32, 10, 10, 10, 10),
              (1, 64, 68, 84, 20, 20),
              (1, 32, 30, 33, 7, 9),      # overlapping windows, token grid not a multiple of the 2x4 block per thread
              (1, 64, 32, 40, 16, 16),    # k (2, 10) s (2, 2): the separable kernel with 4 rows in flight (config 4's P4 level)
              (1, 32, 31, 20, 16, 16),    # k (16, 5) s (1, 1): window taller than the 12 rows in flight -> two chunks
              (16, 64, 40, 40, 16, 16),   # k (10, 10) s (2, 2) with >= 256 workgroups: TWO token rows per workgroup (the default workload's P4 level)
              (32, 32, 30, 33, 7, 9),     # k (6, 9) s (4, 3): two token rows per workgroup, odd token-row count -> a short last block
              (1, 512, 40, 40, 16, 16)]   # yolov5l's P4 level: 512 channels, one token row per workgroup fits only with the packed column maxima


@pytest.mark.parametrize("dt", DTYPES)
def test_axpby_add_fusion(dt):
    x0, x1 = rnd((2, 24, 9, 11), 71), rnd((2, 24, 9, 11), 72)
    a0, a1 = to_act(x0, dt, pad_to=40), to_act(x1, dt)
    y = torch.zeros((2, 9, 11, 24), dtype=dt, device=DEV)
    run(ops.axpby(a0, a1, y, 128.0, -127.0))                  # the reference's Add with weight = channel count
    close(from_act(y), q(x0, dt) * 128.0 + q(x1, dt) * -127.0, dt, "axpby")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", POOL_CASES)
def test_dmff_pool_tokens_and_upsample_merge(case, dt):
    B, C, H, W, va, ha = case
    rgb, ir = rnd((B, C, H, W), 21), rnd((B, C, H, W), 22)
    pos_v, pos_i = rnd((1, va * ha, C), 23, 0.3), rnd((1, va * ha, C), 24, 0.3)
    w_v, w_i = (0.4, 0.7), (0.55, 0.35)
    from icafusion_amd.models.common import AdaptivePool2d
    th, tw, kh, kw, sh, sw = AdaptivePool2d(va, ha).window(H, W)
    ra, ia = to_act(rgb, dt, pad_to=C + 16), to_act(ir, dt)
    tok = torch.zeros((2, B * th * tw, C), dtype=dt, device=DEV)
    run(ops.dmff_pool_tokens(ra, ia, pos_v.reshape(-1).to(DEV), pos_i.reshape(-1).to(DEV), tok, th, tw, kh, kw, sh, sw, w_v, w_i))
    ref_v = oracle.pooled_tokens(q(rgb, dt), va, ha, torch.tensor(w_v[0]), torch.tensor(w_v[1]), pos_v)
    ref_i = oracle.pooled_tokens(q(ir, dt), va, ha, torch.tensor(w_i[0]), torch.tensor(w_i[1]), pos_i)
    close(tok[0].float().cpu().reshape(B, -1, C), ref_v, dt, "pool tokens rgb")
    close(tok[1].float().cpu().reshape(B, -1, C), ref_i, dt, "pool tokens ir")
    # bilinear back-projection + residual + concat, fed with the kernel's own (rounded) tokens
    out = torch.zeros((B, H, W, 2 * C), dtype=dt, device=DEV)
    run(ops.dmff_upsample_merge(tok, ra, ia, out, th, tw))
    tv = tok[0].float().cpu().reshape(B, th, tw, C).permute(0, 3, 1, 2)
    ti = tok[1].float().cpu().reshape(B, th, tw, C).permute(0, 3, 1, 2)
    ref = torch.cat((oracle.bilinear_resize(tv, H, W) + q(rgb, dt), oracle.bilinear_resize(ti, H, W) + q(ir, dt)), 1)
    close(from_act(out), ref, dt, "upsample merge")
    if dt == torch.float32:   # the oracle's bilinear itself equals torch's
        assert torch.allclose(oracle.bilinear_resize(tv, H, W), F.interpolate(tv, size=(H, W), mode="bilinear"), atol=1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [64, 128, 512, 1024])
def test_layernorm(C, dt):
    rows = 37
    x = rnd((2, rows, C), 31, 2.0) + 0.5
    g, b = rnd((2, C), 32, 0.2) + 1.0, rnd((2, C), 33, 0.1)
    xg = x.to(DEV).to(dt)
    y = torch.zeros_like(xg)
    gg, bb = g.to(DEV), b.to(DEV)
    run(ops.layernorm(xg, y, gg[0].contiguous(), bb[0].contiguous(), gg[1].contiguous(), bb[1].contiguous()))
    for k in range(2):
        close(y[k].float().cpu(), oracle.layer_norm(q(x[k], dt), g[k], b[k]), dt, f"layernorm g{k}", factor=2)


ATT_CASES = [(2, 400, 128), (1, 256, 256), (2, 100, 512), (1, 100, 1024), (1, 77, 128), (1, 400, 256), (1, 256, 384),
             (1, 36, 64)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("case", ATT_CASES)
def test_cross_attention(case, dt):
    B, N, C = case
    heads, dk = 8, case[2] // 8
    if dk % ops.VEC[dt]:
        pytest.skip("head dim not a multiple of the 16-byte vector for this dtype")
    qkv = rnd((2, B * N, 3 * C), 41, 1.0)
    qkv[:, :, :2 * C] *= 1.5                      # sharpen the softmax a little
    qg = qkv.to(DEV).to(dt).contiguous()
    out = torch.zeros((2, B * N, C), dtype=dt, device=DEV)
    run(ops.cross_attention(qg, out, B, N, heads))
    f = q(qkv, dt).reshape(2, B, N, 3, heads, dk)
    for d in range(2):
        qq = f[1 - d, :, :, 0].permute(0, 2, 1, 3)          # queries come from the OTHER modality
        kk = f[d, :, :, 1].permute(0, 2, 1, 3)
        vv = f[d, :, :, 2].permute(0, 2, 1, 3)
        att = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(dk), -1)
        ref = (att @ vv).permute(0, 2, 1, 3).reshape(B * N, C)
        close(out[d].float().cpu(), ref, dt, f"cross attention dir {d} {case}", factor=2)


def test_cross_attention_softmax_spike():
    """Force the online-softmax rescale branch: one key dominates late in the key order."""
    B, N, C, heads = 1, 256, 128, 8
    qkv = rnd((2, B * N, 3 * C), 43, 0.3)
    qkv[0, 200, C:2 * C] = 6.0
    qkv[1, :, :C] = qkv[1, :, :C].abs() + 0.5
    qg = qkv.to(DEV).contiguous()
    out = torch.zeros((2, B * N, C), dtype=torch.float32, device=DEV)
    run(ops.cross_attention(qg, out, B, N, heads))
    f = qkv.reshape(2, B, N, 3, heads, 16)
    qq, kk, vv = (f[1, :, :, 0].permute(0, 2, 1, 3), f[0, :, :, 1].permute(0, 2, 1, 3), f[0, :, :, 2].permute(0, 2, 1, 3))
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) / 4.0, -1) @ vv).permute(0, 2, 1, 3).reshape(B * N, C)
    close(out[0].cpu(), ref, torch.float32, "spiked softmax")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [128, 256, 384, 512])
def test_cross_attention_late_spikes_in_16_bit(C, dt):
    """The 16-bit attention loop (attn_core.h) defers the running maximum until a tile exceeds it by 2^6 in the exponent domain, takes the
    softmax denominator out of the matrix pipe (d_k = 16 / 48: a ones row of V^T; d_k = 32: an extra MFMA) and rescales O — and with it
    the denominator — on the cold path.  Keys that dominate LATE in the key order, by less and by much more than the deferral slack,
    in the last (partly padded) key tile as well, exercise every one of those paths (d_k = 16, 32, 48, 64)."""
    B, N, heads = 2, 300, 8
    dk = C // heads
    qkv = rnd((2, B * N, 3 * C), 47, 0.4)
    qkv[:, :, :C] = qkv[:, :, :C].abs() + 0.3                     # positive queries: a large positive key raises every score of its column
    for key, amp in ((40, 1.2), (131, 3.0), (222, 7.0), (297, 12.0)):   # growing spikes: below the slack, around it, far above it
        qkv[0, key, C:2 * C] = amp
        qkv[1, N + key, C:2 * C] = amp * 0.9
    qg = qkv.to(DEV).to(dt).contiguous()
    out = torch.zeros((2, B * N, C), dtype=dt, device=DEV)
    run(ops.cross_attention(qg, out, B, N, heads))
    f = q(qkv, dt).reshape(2, B, N, 3, heads, dk)
    for d in range(2):
        qq, kk, vv = (f[1 - d, :, :, 0].permute(0, 2, 1, 3), f[d, :, :, 1].permute(0, 2, 1, 3), f[d, :, :, 2].permute(0, 2, 1, 3))
        ref = (torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(dk), -1) @ vv).permute(0, 2, 1, 3).reshape(B * N, C)
        close(out[d].float().cpu(), ref, dt, f"late spikes dir {d} C={C}", factor=2)


@pytest.mark.parametrize("path", ["pixel", "element"])
@pytest.mark.parametrize("nc", [1, 3, 9])
def test_detect_decode(nc, path, monkeypatch):
    """Both kernels behind icaf_detect_decode: one thread per pixel (3 anchors, no = 6 / 8 / 14) and the general one thread per element."""
    if path == "element":
        monkeypatch.setenv("ICAF_DETECT_ELEMENTWISE", "1")
    B, na, no = 2, 3, nc + 5
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    dims = [(12, 20), (6, 10), (3, 5)]
    rows = sum(na * h * w for h, w in dims)
    z = torch.zeros((B, rows, no), dtype=torch.float32, device=DEV)
    lg = torch.zeros((B, rows, nc), dtype=torch.float32, device=DEV)
    feats, raws, off = [], [], 0
    for l, (h, w) in enumerate(dims):
        p = rnd((B, na * no, h, w), 50 + l, 2.0)
        feats.append(p)
        pa = torch.zeros((B, h, w, na * no + (2 if l else 0)), dtype=torch.float32, device=DEV)     # dense (as the plan allocates it) / padded pixel stride
        pa[..., :na * no] = p.permute(0, 2, 3, 1).to(DEV)
        raw = torch.zeros((B, na, h, w, no), dtype=torch.float32, device=DEV)
        run(ops.detect_decode(pa[..., :na * no], z, lg, raw, na, no, off, oracle.STRIDES[l], anchors[l]))
        raws.append(raw)
        off += na * h * w
    # oracle: identity "conv" weights so detect() sees the same maps
    sd = {}
    for l in range(3):
        sd[f"d.m.{l}.weight"] = torch.eye(na * no).reshape(na * no, na * no, 1, 1)
        sd[f"d.m.{l}.bias"] = torch.zeros(na * no)
    rz, rl, rr = oracle.detect(feats, sd, "d", nc, anchors)
    assert torch.allclose(z.cpu(), rz, rtol=1e-6, atol=1e-5)
    assert torch.equal(lg.cpu(), rl)
    for a, b in zip(raws, rr):
        assert torch.equal(a.cpu(), b)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nc", [1, 3, 9])
def test_detect_level_in_one_launch(nc, dt):
    """icaf_detect_conv (the level's 1x1 conv as the persistent streaming GEMM with the decode as its epilogue) must give the SAME
    BITS — z, logits, raw — as icaf_conv2d (fp32 out) followed by icaf_detect_decode, on every level shape of a plan: many pixel
    tiles per workgroup (80 x 80 x 6 images), fewer tiles than workgroups (20 x 20), a ragged last tile."""
    B, na, no = 6, 3, nc + 5
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    dims = [(80, 80, 128), (37, 41, 256), (20, 20, 512)]
    rows = sum(na * h * w for h, w, _ in dims)
    zs = [torch.zeros((B, rows, no), dtype=torch.float32, device=DEV) for _ in range(2)]
    lgs = [torch.zeros((B, rows, nc), dtype=torch.float32, device=DEV) for _ in range(2)]
    raws, off = [[], []], 0
    for l, (h, w, c) in enumerate(dims):
        x = to_act(rnd((B, c, h, w), 70 + l), dt, pad_to=c + 16)
        wt = rnd((na * no, c, 1, 1), 80 + l, 2.0 / math.sqrt(c))
        bias = rnd((na * no,), 90 + l, 1.0)
        wp, kp = ops.pack_conv_weight(wt.to(DEV), dt)
        bp = ops.pack_bias(bias.to(DEV), na * no)
        assert ops.detect_conv_ok(x, na, no, c)
        r1 = torch.zeros((B, na, h, w, no), dtype=torch.float32, device=DEV)
        run(ops.detect_conv(x, wp, kp, bp, zs[0], lgs[0], r1, na, no, off, oracle.STRIDES[l], anchors[l], c))
        pmap = torch.zeros((B, h, w, (na * no + 3) // 4 * 4), dtype=torch.float32, device=DEV)[..., :na * no]
        r2 = torch.zeros_like(r1)
        run(ops.conv2d(x, wp, kp, bp, pmap, 1, 1, 1, 1, 0, 0, c, na * no, ops.ACT_NONE))
        run(ops.detect_decode(pmap, zs[1], lgs[1], r2, na, no, off, oracle.STRIDES[l], anchors[l]))
        raws[0].append(r1); raws[1].append(r2)
        off += na * h * w
    assert torch.equal(zs[0], zs[1]) and torch.equal(lgs[0], lgs[1])
    assert all(torch.equal(a, b) for a, b in zip(*raws))
    assert float(zs[0].abs().max()) > 1.0 and float(lgs[0].abs().max()) > 0.0


def _rand_pred(B, rows, nc, seed, ties=False):
    g = np.random.default_rng(seed)
    xy = g.uniform(0, 640, (B, rows, 2))
    wh = g.uniform(4, 220, (B, rows, 2))
    obj = g.uniform(0, 1, (B, rows, 1))
    cls = g.uniform(0, 1, (B, rows, nc))
    p = np.concatenate((xy, wh, obj, cls), 2).astype(np.float32)
    if ties:
        k = 350 if ties is True else int(ties)
        p[:, 50:50 + k, 4:] = p[:, 50:51, 4:]        # many identical scores: order must fall back to the index
    return p


NMS_CASES = [
    dict(B=2, rows=3000, nc=1, conf=0.25, iou=0.45),
    dict(B=3, rows=25200, nc=1, conf=0.001, iou=0.5),
    dict(B=2, rows=5000, nc=9, conf=0.2, iou=0.5, multi_label=True),
    dict(B=1, rows=6300, nc=9, conf=0.05, iou=0.6, multi_label=True),        # > max_nms candidates
    dict(B=2, rows=2000, nc=4, conf=0.3, iou=0.5, agnostic=True, classes=[0, 2]),
    dict(B=2, rows=1500, nc=1, conf=0.1, iou=0.5, ties=True),
    dict(B=2, rows=800, nc=3, conf=0.999, iou=0.5),                           # nothing survives
    dict(B=1, rows=37, nc=2, conf=0.05, iou=0.3, multi_label=True),
    # the select / gather / sort / walk rounds of nms_walk_kernel (nms.hip):
    dict(B=2, rows=25200, nc=1, conf=0.001, iou=0.05),                        # aggressive suppression: the walk needs many rounds
    dict(B=2, rows=9000, nc=1, conf=0.05, iou=0.5, ties=6000),                # 6000 identical scores: one histogram bin > 4096 keys,
    #                                                                           refined down to the slot-index digits
    dict(B=2, rows=12000, nc=1, conf=0.05, iou=0.3, near=True),               # 12000 distinct scores inside ONE bin (upper 16 bits equal)
    dict(B=1, rows=20000, nc=3, conf=0.01, iou=0.2, multi_label=True, ties=9000),   # > max_nms candidates AND a refined bin
    dict(B=3, rows=1, nc=1, conf=0.1, iou=0.5),                               # a single row
]


@pytest.mark.parametrize("case", NMS_CASES)
def test_nms_bit_exact(case):
    from icafusion_amd.utils.general import nms_device, non_max_suppression
    c = dict(case)
    B, rows, nc = c.pop("B"), c.pop("rows"), c.pop("nc")
    conf, iou, ties, near = c.pop("conf"), c.pop("iou"), c.pop("ties", False), c.pop("near", False)
    pred = _rand_pred(B, rows, nc, seed=rows + nc, ties=ties)
    if near:                                         # scores in [0.5, 0.5 + 2^-9): same exponent and upper mantissa bits
        g = np.random.default_rng(9)
        pred[..., 4] = 1.0
        pred[..., 5:] = (0.5 + g.uniform(0, 2.0 ** -9, pred[..., 5:].shape)).astype(np.float32)
    ref, ref_idx = oracle.non_max_suppression(pred, conf, iou, return_indices=True, **c)
    pt = torch.from_numpy(pred).to(DEV)
    got = non_max_suppression(pt, conf, iou, **c)
    det, count, keep = nms_device(pt, conf, iou, **c)
    for b in range(B):
        np.testing.assert_array_equal(got[b].cpu().numpy(), ref[b])
        n = int(count[b])
        assert n == len(ref_idx[b])
        np.testing.assert_array_equal(keep[b, :n].cpu().numpy().astype(np.int64), ref_idx[b])


@pytest.mark.parametrize("name,src", [("nms_s_conf25", "model_s_kaist_320_b2"), ("nms_s_conf30", "model_s_kaist_320_b2"),
                                      ("nms_s_conf001_iou5", "model_s_kaist_320_b2"),
                                      ("nms_l_multilabel", "model_l_vedai_320_b1"),
                                      ("nms_l_agnostic_classes", "model_l_vedai_320_b1")])
def test_nms_golden_fixtures(name, src):
    """Against the committed outputs of the reference's own non_max_suppression wrapper."""
    from helpers import load_golden
    from icafusion_amd.utils.general import non_max_suppression
    g, z = load_golden(name), load_golden(src)["z"]
    kw = dict(eval(str(g["kw"])))
    got = non_max_suppression(torch.from_numpy(z).to(DEV), **kw)
    for i, o in enumerate(got):
        np.testing.assert_array_equal(o.cpu().numpy(), g[f"det{i}"])


def test_graph_capture_replays_conv():
    dt = torch.bfloat16
    x = rnd((1, 64, 16, 16), 61)
    w = rnd((64, 64, 3, 3), 62, 0.05)
    plan = Plan(DEV, dt)
    xa = to_act(x, dt)
    wp, kp = ops.pack_conv_weight(w.to(DEV), dt)
    y = plan.act(1, 16, 16, 64)
    plan.add(ops.conv2d(xa, wp, kp, None, y, 3, 3, 1, 1, 1, 1, 64, 64, ops.ACT_SILU))
    plan.run(); torch.cuda.synchronize()
    eager = y.clone()
    y.zero_()
    plan.capture()
    y.zero_()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.run(s.cuda_stream)
    s.synchronize()
    assert torch.equal(y, eager)


@pytest.mark.parametrize("seed,with_scale", [(1, False), (2, True), (3, True)])
def test_match_predictions_equals_host_statement(seed, with_scale):
    """icaf_match_predictions (scale_coords + clip + per-class best-IoU matching of test.py:196-230 on the device) vs the
    numpy statement in icafusion_amd/utils/metrics.py: identical TP flags, identical native-space boxes."""
    from icafusion_amd.utils.metrics import match_predictions as host_match
    g = np.random.default_rng(seed)
    B, max_det, nc = 5, 300, 4
    det = np.zeros((B, max_det, 6), np.float32)
    count = g.integers(0, max_det + 1, B).astype(np.int32)
    count[0] = 0
    labels, off, scales = [], [0], []
    for b in range(B):
        n = count[b]
        xy = g.uniform(0, 600, (n, 2)); wh = g.uniform(5, 120, (n, 2))
        det[b, :n, :2], det[b, :n, 2:4] = xy, xy + wh
        det[b, :n, 4], det[b, :n, 5] = np.sort(g.uniform(0.1, 1, n))[::-1], g.integers(0, nc, n)
        nl = 0 if b == 1 else int(g.integers(1, 40))
        gain, px, py = (float(g.uniform(0.4, 1.6)), float(g.uniform(0, 40)), float(g.uniform(0, 40))) if with_scale else (1.0, 0.0, 0.0)
        w0, h0 = (500.0, 420.0) if with_scale else (3.0e38, 3.0e38)
        scales.append([gain, px, py, w0, h0])
        lab = np.zeros((nl, 5), np.float32)
        for k in range(nl):                       # labels near detections (so that matches exist), in native space
            if n and g.random() < 0.7:
                src = det[b, g.integers(0, n)]
                box = (src[:4] - np.array([px, py, px, py], np.float32)) / np.float32(gain) + g.normal(0, 4, 4)
                lab[k] = [src[5], *box]
            else:
                x, y = g.uniform(0, 400, 2); lab[k] = [g.integers(0, nc), x, y, x + g.uniform(5, 90), y + g.uniform(5, 90)]
        labels.append(lab); off.append(off[-1] + nl)
    lab_all = np.concatenate(labels, 0).astype(np.float32)
    iouv = np.linspace(0.5, 0.95, 10).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    scale_t = dev(np.array(scales, np.float32)) if with_scale else None
    predn = torch.zeros((B, max_det, 4), dtype=torch.float32, device=DEV)
    correct = ops.match_predictions(dev(det), dev(count), dev(lab_all), dev(np.array(off, np.int32)), dev(iouv), scale=scale_t, predn=predn)
    torch.cuda.synchronize()
    correct, predn = correct.cpu().numpy().astype(bool), predn.cpu().numpy()
    for b in range(B):
        n = count[b]
        pn = det[b, :n].copy()
        if with_scale:
            gain, px, py, w0, h0 = (np.float32(v) for v in scales[b])
            pn[:, [0, 2]] = np.clip((pn[:, [0, 2]] - px) / gain, 0, w0)
            pn[:, [1, 3]] = np.clip((pn[:, [1, 3]] - py) / gain, 0, h0)
        np.testing.assert_array_equal(predn[b, :n], pn[:, :4])
        ref = host_match(pn, labels[b], iouv)
        np.testing.assert_array_equal(correct[b, :n], ref, err_msg=f"image {b}")


def test_match_predictions_kernel_reproduces_the_reference_block():
    """icaf_match_predictions fed the letterboxed NMS rows + scale parameters vs tests/golden/match_predictions.npz (the
    reference's inline block test.py:196-230 exec'ed by make_golden.py): same native-space boxes, same TP flags."""
    from helpers import load_golden
    g = load_golden("match_predictions")
    T, max_det = int(g["n"]), 300
    iouv = g["iouv"].astype(np.float32)
    det = np.zeros((T, max_det, 6), np.float32)
    count = np.zeros(T, np.int32)
    labs, off, scales = [], [0], []
    for t in range(T):
        pred, tbox, labels = g[f"pred{t}"], g[f"tbox{t}"], g[f"labels{t}"]
        H, W, h0, w0, gain, pw, ph = g[f"geom{t}"]
        count[t] = len(pred)
        det[t, :len(pred)] = pred
        labs.append(np.concatenate((labels[:, :1], tbox), 1).astype(np.float32) if len(labels) else np.zeros((0, 5), np.float32))
        off.append(off[-1] + len(labels))
        scales.append([gain, pw, ph, w0, h0])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    predn = torch.zeros((T, max_det, 4), dtype=torch.float32, device=DEV)
    correct = ops.match_predictions(dev(det), dev(count), dev(np.concatenate(labs, 0)), dev(np.array(off, np.int32)), dev(iouv),
                                    scale=dev(np.array(scales, np.float32)), predn=predn)
    torch.cuda.synchronize()
    correct, predn = correct.cpu().numpy().astype(bool), predn.cpu().numpy()
    for t in range(T):
        n = count[t]
        np.testing.assert_array_equal(predn[t, :n], g[f"predn{t}"][:, :4], err_msg=f"boxes, trial {t}")
        if len(labs[t]):
            np.testing.assert_array_equal(correct[t, :n], g[f"correct{t}"], err_msg=f"flags, trial {t}")


def test_nms_greedy_core_against_torchvision_if_the_box_has_it():
    """The one parity-unpinned piece: the greedy core of non_max_suppression lives in torchvision.ops.nms
    (utils/general.py:591; torchvision>=0.8.1, un-vendored), which the build container does not have.  If THIS box has it, the CPU
    restatement (oracle.nms_greedy) is cross-checked against it on 60 random configurations (clustered boxes, ties, degenerate boxes,
    thresholds 0.3-0.7) — kept indices equal, in order.  Either way the finding is written down: gpurun_out/torchvision_probe.json
    (copied to profiles/ by hand) says whether the pin exists on the GPU box."""
    import json
    import os
    rec = {"torchvision_present": False, "version": None, "configurations_checked": 0, "all_equal": None}
    try:
        import torchvision                                          # noqa: F401
        from torchvision.ops import nms as tv_nms
        rec.update(torchvision_present=True, version=torchvision.__version__)
    except Exception as e:                                          # absent (or unusable) on this image: recorded, then skipped (below)
        rec["import_error"] = f"{type(e).__name__}: {e}"[:200]
        tv_nms = None
    if tv_nms is not None:
        g = np.random.default_rng(2024)
        ok = True
        for k in range(60):
            n = int(g.integers(1, 400))
            ctr = g.uniform(0, 200, (max(1, n // 8), 2))
            c = ctr[g.integers(0, len(ctr), n)] + g.normal(0, 6, (n, 2))
            wh = g.uniform(0, 40, (n, 2)) * (g.uniform(size=(n, 1)) > 0.03)          # a few zero-area boxes
            boxes = np.concatenate((c - wh / 2, c + wh / 2), 1).astype(np.float32)
            scores = g.uniform(size=n).astype(np.float32)
            if k % 3 == 0:
                scores = np.round(scores * 20) / 20                                  # ties
            thr = float(g.choice([0.3, 0.45, 0.5, 0.6, 0.7]))
            want = tv_nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
            got = oracle.nms_greedy(boxes, scores, thr)
            # torchvision sorts with an unstable sort: among EQUAL scores its visiting order is unspecified, so tied configurations are
            # compared as sets of kept boxes per score value; distinct scores must match index for index
            if len(np.unique(scores)) == n:
                ok = ok and np.array_equal(got, want)
            else:
                ok = ok and sorted(scores[got].tolist()) == sorted(scores[want].tolist())
            rec["configurations_checked"] += 1
        rec["all_equal"] = bool(ok)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "torchvision_probe.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    if tv_nms is None:                                              # nothing was compared: a SKIP in the record of the run, not a pass
        pytest.skip(f"torchvision is absent on this box ({rec.get('import_error')}): the greedy core stays parity-unpinned (finding written to {out}/torchvision_probe.json)")
    assert rec["all_equal"] is True
