#!/usr/bin/env python3
"""Phase clocks of cwide_kernel (build: python tools/quick_variant.py cwdbg cwide.hip -DICAF_CW_DBG; run with ICAF_LIB=.../libicaf_cwdbg.so):
first and last workgroup of a 3x3 128 -> 128 layer at 40 x 40, batch 64 (both backbones)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from icafusion_amd import ops
from icafusion_amd._lib import lib

DEV = "cuda:0"
dt = torch.bfloat16
G, B, H, W, c = 2, 32, 40, 40, 128
x = torch.randn(G, B, H, W, c, device=DEV).to(dt)
r = torch.randn(G, B, H, W, c, device=DEV).to(dt)
ws = [torch.randn(c, c, 3, 3) / math.sqrt(c * 9) for _ in range(G)]
packs = [ops.pack_conv_weight(w.to(DEV), dt) for w in ws]
wp, kp = torch.stack([p[0] for p in packs]).contiguous(), packs[0][1]
bp = torch.stack([ops.pack_bias(torch.zeros(c, device=DEV), c) for _ in range(G)]).contiguous()
y = torch.zeros(G, B, H, W, c, dtype=dt, device=DEV)
for tile in (81, 82):
    l = ops.conv2d(x, wp, kp, bp, y, 3, 3, 1, 1, 1, 1, c, c, ops.ACT_SILU, res=r, tile=tile)
    sp = ops.current_stream_ptr()
    for _ in range(3): l(sp)
    torch.cuda.synchronize()
    e0, e1 = ops.Event(), ops.Event(); e0.record(sp); [l(sp) for _ in range(20)]; e1.record(sp); torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    assert lib().icaf_cwide_debug_clocks(out) == 0
    s = list(out)
    names = ["issue", "wait patch", "K loop", "barrier", "stage", "flush"]
    for k, base in (("first wg", 0), ("last wg", 16)):
        d = [s[base + i + 1] - s[base + i] for i in range(6)]
        pro = f"prologue: weights issued +{s[base + 8] - s[base]}, patch DMA issued +{s[base + 9] - s[base + 8]}, residual + bias loads issued +{s[base + 10] - s[base + 9]}, to the wait +{s[base + 1] - s[base + 10]}"
        print(f"tile {tile} {k}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d)) + f"  | total {s[base + 6] - s[base]} cycles | {pro}")
    print(f"tile {tile}: kernel {e0.elapsed_ms(e1) / 20 * 1e3:.1f} us (s_memtime counts shader-clock cycles here)")
