"""Every launch configuration (tile x pipeline) of every conv launch of a plan must produce bit-identical output: run each
candidate on the launch's real input and compare.  python lab/probes/tile_invariance.py [s|l] [batch]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.synth import synth_images, synth_state_dict
mdl = sys.argv[1] if len(sys.argv) > 1 else "s"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5{mdl}_Transfusion_kaist.yaml"))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 3)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
m.autotune = False; m.use_graph = False
plan = m.plan_for(B, 640, 640, "cuda:0")
rgb, ir = synth_images(B, 640, 640, seed=3)
plan.inputs[0].copy_(rgb.cuda()); plan.inputs[1].copy_(ir.cuda())
sp = ops.current_stream_ptr()
bad = 0
snaps = {}
for i, l in enumerate(plan.launches):
    if l.fn is ops.lib().icaf_conv2d and l.keep[0].res and l.keep[0].res == l.keep[0].y:
        snaps[i] = l.keep[4].clone()
    l(sp); torch.cuda.synchronize()
    if l.fn is not ops.lib().icaf_conv2d:
        continue
    a = l.keep[0]
    y = l.keep[4]; chain = l.keep[7]
    outs = [y] + ([chain["y"]] if chain else [])
    inplace = bool(a.res) and a.res == a.y     # in place over its residual: restore the buffer before every run
    ref, keep = None, a.tile
    if inplace:
        snap = snaps[i]
    for c in ops.conv_candidates(a):
        a.tile = c
        if inplace:
            y.copy_(snap)
        if l.fn(*l.args, sp) != 0:
            continue
        torch.cuda.synchronize()
        got = [o.clone() for o in outs]
        if ref is None:
            ref, rc = got, c
        elif not all(torch.equal(g, r) for g, r in zip(got, ref)):
            d = max((g.float() - r.float()).abs().max().item() for g, r in zip(got, ref))
            print(f"launch {i} {l.name}: tile {c} differs from tile {rc} (max |d| = {d:.3e}) M={a.B * a.Ho * a.Wo} N={a.Cout} K={a.kh * a.kw * a.Cin} pre={bool(a.pre)} out_dtype={a.out_dtype} act={a.act}")
            bad += 1
    a.tile = keep
    if inplace:
        y.copy_(snaps[i])
    l(sp); torch.cuda.synchronize()
print("mismatching (launch, tile) pairs:", bad)
