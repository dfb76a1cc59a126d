"""Time icaf_stem2 alone (batch 32, 640x640, both streams) — with a debug build (tools/build_variant.py + ICAF_STEM2_DBG bits:
1 no prefetch/commit, 2 no stage 1, 4 no stage 2, 8 no stage 3, 16 no stores) this gives the per-stage cost."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch
from icafusion_amd import ops
dt = torch.bfloat16; DEV = "cuda:0"
img = torch.rand((2, 32, 3, 640, 640), device=DEV)
mk = lambda shape: torch.randn(shape, device=DEV) * 0.05
p0 = [ops.pack_conv_weight(ops.s2d_conv_weight(mk((32, 3, 6, 6))), dt, 16) for _ in range(2)]
p1 = [ops.pack_conv_weight(mk((64, 32, 3, 3)), dt) for _ in range(2)]
p2 = [ops.pack_conv_weight(mk((64, 64, 1, 1)), dt) for _ in range(2)]
st = lambda ts: torch.stack(ts).contiguous()
w0, w1, w2 = (st([p[0] for p in ps]) for ps in (p0, p1, p2))
b0, b1, b2 = (st([ops.pack_bias(mk((n,)), n) for _ in range(2)]) for n in (32, 64, 64))
y = torch.zeros((2, 32, 160, 160, 96), dtype=dt, device=DEV)[..., :64]
l = ops.stem2(img, w0, p0[0][1], b0, w1, p1[0][1], b1, w2, p2[0][1], b2, y, 32, 64, 64)
sp = ops.current_stream_ptr()
for _ in range(3): l(sp)
torch.cuda.synchronize()
e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
for _ in range(20): l(sp)
e1.record(sp); torch.cuda.synchronize()
print(f"dbg={os.environ.get('ICAF_STEM2_DBG', '0'):>3s}: {e0.elapsed_ms(e1) / 20 * 1e3:7.1f} us")
