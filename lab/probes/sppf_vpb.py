#!/usr/bin/env python3
"""GPU probe: SPPF pooling launch (icaf_sppf_pool, LDS version) with the channel-vector group per workgroup capped at 8 / 4 / 2 / 1
(ICAF_SPPF_VPB) — fewer vectors per workgroup = more, smaller workgroups (latency hiding) against shorter contiguous runs per pixel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch   # noqa: E402

from icafusion_amd import ops   # noqa: E402

dev = "cuda:0"
for (B, H, W, C, dt) in ((64, 20, 20, 256, torch.bfloat16), (64, 20, 20, 512, torch.bfloat16), (32, 40, 40, 512, torch.float16)):
    x = torch.randn(B, H, W, C, device=dev).to(dt)
    ys = [torch.empty_like(x) for _ in range(3)]
    ref = None
    for cap in (8, 4, 2, 1):
        os.environ["ICAF_SPPF_VPB"] = str(cap)
        l = ops.sppf_pool(x, *ys, 5)
        sp = ops.current_stream_ptr()
        for _ in range(5):
            l(sp)
        torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(sp)
        for _ in range(50):
            l(sp)
        e1.record(sp)
        us = e0.elapsed_ms(e1) * 1e3 / 50
        out = torch.stack([y.float() for y in ys])
        ref = out if ref is None else ref
        print(f"B={B} {H}x{W} C={C} {str(dt)[6:]} cap={cap}: {us:7.1f} us  {4 * x.numel() * x.element_size() / us / 1e3:6.0f} GB/s  same={bool(torch.equal(out, ref))}")
