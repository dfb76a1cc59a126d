#!/usr/bin/env python3
"""Phase breakdown of icaf_dmff_attn_mlp (shader-clock stamps of workgroup (0,0,0)): staging of the first K / V^T pair, attention
(4 rounds incl. staging), out-projection, LayerNorm, MLP, output.  python lab/probes/dmff_phases.py  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from icafusion_amd import ops
from icafusion_amd.engine import Plan
from icafusion_amd.models.common import CrossTransformerBlock
from icafusion_amd.synth import synth_tensor

for C, N, B in ((128, 400, 32), (256, 256, 32), (512, 100, 32)):
    blk = CrossTransformerBlock(C, C, C, 8, 4, 0.1, 0.1).eval()
    blk.load_state_dict({k: synth_tensor("b." + k, v.shape, seed=1) for k, v in blk.state_dict().items()})
    blk = blk.to("cuda:0"); blk.fuse_block = True; blk.fuse_max_c = 512
    plan = Plan("cuda:0", torch.bfloat16)
    t = plan.tokens(2, B * N, C); t.copy_(torch.randn(2, B * N, C, device="cuda:0").to(torch.bfloat16))
    blk.emit_tokens(plan, t, B, N)
    dbg = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    l = plan.launches[1]; l.keep[0].debug_clock = dbg.data_ptr()
    for _ in range(3): plan.run()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy(); d = (d[1:7] - d[0:6])
    e0, e1 = ops.Event(), ops.Event(); sp = ops.current_stream_ptr()
    e0.record(sp); [l(sp) for _ in range(10)]; e1.record(sp); ms = e0.elapsed_ms(e1) / 10
    print(f"C={C} N={N} B={B}: kernel {ms*1e3:.1f} us | clocks: stage0 {d[0]} attention(rest) {d[1]} out-proj {d[2]} LN {d[3]} MLP {d[4]} store {d[5]} | total {d.sum()} ({d.sum()/2.4e3:.1f} us @2.4GHz)")
