"""s_memtime stamps of workgroup 0 of the producer / consumer stem2 (probe build: python tools/quick_variant.py s2pcclk stem.hip -DICAF_S2_PC=1 -DICAF_S2_CLK=1):
0 loop top, 1 commit + prefetch issued, 2 barrier (a), [consumers: 3 stage-2 MFMAs done, 4 epilogue 1 done, 5 stage 3 + epilogue 2 done], 6 work done, 7 barrier (b)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch
DEV = "cuda:0"
NT, NS = 16, 12
clk = torch.zeros((8, NT, NS), dtype=torch.int64, device=DEV)
os.environ["ICAF_S2_CLK_PTR"] = hex(clk.data_ptr())
from icafusion_amd import ops
dt = torch.bfloat16
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
for _ in range(10): l(sp)
e1.record(sp); torch.cuda.synchronize()
print(f"stem2 producer/consumer (probe build): {e0.elapsed_ms(e1) / 10 * 1e3:7.1f} us")
c = clk.cpu().numpy()
for w in range(8):
    T = slice(4, 12)
    tile = (c[w, 5:13, 0] - c[w, 4:12, 0]).mean()
    seg = lambda a, b: (c[w, T, b] - c[w, T, a]).mean()
    if w < 4:
        print(f"producer {w}: tile {tile:7.0f} | commit+fetch {seg(0, 1):5.0f} | wait (a) {seg(1, 2):5.0f} | stage 1 {seg(2, 6):5.0f} | wait (b) {seg(6, 7):5.0f}")
    else:
        print(f"consumer {w - 4}: tile {tile:7.0f} | commit+fetch {seg(0, 1):5.0f} | wait (a) {seg(1, 2):5.0f} | stage-2 MFMAs {seg(2, 3):5.0f} | epilogue {seg(3, 4):5.0f} | "
              f"stage 3 {seg(4, 5):5.0f} | stores {seg(5, 6):5.0f} | wait (b) {seg(6, 7):5.0f}")
