#!/usr/bin/env python3
"""Per-launch times of one CrossTransformerBlock iteration at the three yolov5s levels (batch 32, bf16), for each launch structure:
per-layer (7), two launches (ln_qkv + attn_mlp), three launches (ln_qkv + attention + proj_mlp).  python lab/probes/dmff_levels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from icafusion_amd import ops
from icafusion_amd.engine import Plan
from icafusion_amd.models.common import CrossTransformerBlock
from icafusion_amd.synth import synth_tensor

ops.load_tune_cache(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", "tune_cache.json"))
LEVELS = {"s": ((128, 400, 32), (256, 256, 32), (512, 100, 32)), "l": ((256, 400, 32), (512, 256, 32), (1024, 100, 32))}
for C, N, B in LEVELS[sys.argv[1] if len(sys.argv) > 1 else "s"]:
    blk = CrossTransformerBlock(C, C, C, 8, 4, 0.1, 0.1).eval()
    blk.load_state_dict({k: synth_tensor("b." + k, v.shape, seed=1) for k, v in blk.state_dict().items()})
    blk = blk.to("cuda:0")
    for mode, (fb, mc, wide) in (("per-layer", (False, 128, False)), ("two", (True, 512, False)), ("three", (True, 64, True))):
        if (mode == "three" and not ops.dmff_wide_ok(C, 4 * C, torch.bfloat16)) or (mode == "two" and C > 512):
            continue
        blk.fuse_block, blk.fuse_max_c, blk.fuse_wide = fb, mc, wide
        blk.invalidate()
        plan = Plan("cuda:0", torch.bfloat16)
        t = plan.tokens(2, B * N, C); t.copy_(torch.randn(2, B * N, C, device="cuda:0").to(torch.bfloat16))
        blk.emit_tokens(plan, t, B, N)
        for _ in range(3): plan.run()
        torch.cuda.synchronize()
        sp = ops.current_stream_ptr()
        per = []
        for l in plan.launches:
            e0, e1 = ops.Event(), ops.Event()
            e0.record(sp); [l(sp) for _ in range(20)]; e1.record(sp); torch.cuda.synchronize()
            per.append((l.name, e0.elapsed_ms(e1) / 20 * 1e3))
        e0, e1 = ops.Event(), ops.Event()
        e0.record(sp); [plan.run() for _ in range(20)]; e1.record(sp); torch.cuda.synchronize()
        print(f"C={C} N={N} {mode:9s}: block {e0.elapsed_ms(e1) / 20 * 1e3:6.1f} us | " + " ".join(f"{n}={u:.1f}" for n, u in per))
