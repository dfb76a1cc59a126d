"""In-situ timing of every launch configuration of the chained conv launches of the bench plan (yolov5s bf16 b32 640)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.synth import synth_state_dict
cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5{sys.argv[1] if len(sys.argv) > 1 else 's'}_Transfusion_kaist.yaml"))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
m.autotune = True; m.use_graph = False
ops.load_tune_cache(f"{R}/profiles/tune_cache.json")
plan = m.plan_for(32, 640, 640, "cuda:0")
sp = ops.current_stream_ptr()
plan.run(sp); torch.cuda.synchronize()
for i, l in enumerate(plan.launches):
    if l.fn is not ops.lib().icaf_conv2d or not l.keep[0].w2:
        continue
    a = l.keep[0]
    keep = a.tile
    res = []
    for c in ops.conv_candidates(a):
        a.tile = c
        if l.fn(*l.args, sp) != 0:
            res.append((c, None)); continue
        e0, e1 = ops.Event(), ops.Event(); ms = 0.0
        for _ in range(5):
            for k in plan.launches[max(0, i - 6):i]:
                k(sp)
            e0.record(sp); l.fn(*l.args, sp); e1.record(sp); torch.cuda.synchronize()
            ms += e0.elapsed_ms(e1)
        res.append((c, ms / 5 * 1e3))
    a.tile = keep
    print(i, l.name, f"M={a.B * a.Ho * a.Wo} N={a.Cout} K={a.kh * a.kw * a.Cin} N2={a.Cout2} cached={keep}:", " ".join(f"{c}={'n/a' if t is None else f'{t:.1f}'}" for c, t in res), flush=True)
