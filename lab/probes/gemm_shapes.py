#!/usr/bin/env python3
"""How close does the production igemm kernel get to the steady-state probe?  Plain GEMMs (1x1 conv) of growing K at the
M / N of the 40x40 layers, and the same K as a 3x3 convolution, on tile 128x128 (pipeline 2) and 256x256."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from icafusion_amd import ops
dev, dt = "cuda:0", torch.bfloat16
sp = ops.current_stream_ptr()
def timeit(l, reps=10):
    l(sp); torch.cuda.synchronize()
    e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
    for _ in range(reps): l(sp)
    e1.record(sp); return e0.elapsed_ms(e1) / reps
for (B, H, W, cin, cout, k) in [(64, 40, 40, 128, 128, 1), (64, 40, 40, 1152, 128, 1), (64, 40, 40, 8192, 128, 1), (64, 40, 40, 128, 128, 3),
                                (256, 40, 40, 128, 128, 3), (64, 40, 40, 2048, 256, 1), (64, 40, 40, 256, 256, 3), (512, 40, 40, 1152, 128, 1)]:
    x = torch.randn((B, H, W, cin), device=dev).to(dt)
    w = torch.randn((cout, cin, k, k), device=dev) * 0.02
    wp, kp = ops.pack_conv_weight(w, dt)
    y = torch.zeros((B, H, W, cout), dtype=dt, device=dev)
    res = []
    for tile in (21, 22, 25, 26, 1, 31):
        if tile == 26 and cout < 256: continue
        try:
            l = ops.conv2d(x, wp, kp, None, y, k, k, 1, 1, k // 2, k // 2, cin, cout, ops.ACT_SILU, tile=tile)
            ms = timeit(l)
            res.append(f"t{tile}: {ms*1e3:7.1f}us {l.flops/ms/1e9:7.1f}TF")
        except Exception as e:
            res.append(f"t{tile}: n/a")
    print(f"M={B*H*W:7d} N={cout:4d} K={cin*k*k:5d} {k}x{k}  " + "  ".join(res))
