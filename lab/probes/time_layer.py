#!/usr/bin/env python3
"""Time single conv launches of the bench plan (yolov5s — ICAF_PROBE_MODEL=l: yolov5l — bf16, batch 32, 640x640) under given launch configurations:
    python lab/probes/time_layer.py idx:tile[,tile...] [idx:tile,...] ...        (ICAF_LIB selects the library build)
prints microseconds per launch (median of 5 x 10 back-to-back launches) - for same-box A/B of kernel variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.synth import synth_state_dict
MODEL = os.environ.get("ICAF_PROBE_MODEL", "s")          # "l": the yolov5l shard (config 3)
cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", f"yolov5{MODEL}_Transfusion_kaist.yaml")))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
if os.environ.get("ICAF_PROBE_NOTUNE"):                 # forced configurations only: skip the tuner (a plan build is then a few seconds)
    m.autotune = False
plan = m.plan_for(32, 640, 640, "cuda:0")
plan.run(); torch.cuda.synchronize()
sp = ops.current_stream_ptr()
out = {}
for spec in sys.argv[1:]:
    idx, tiles = spec.split(":")
    l = plan.launches[int(idx)]
    for t in tiles.split(","):
        l.keep[0].tile = int(t)
        if l.fn(*l.args, sp) != 0:
            out[f"{idx}:{t}"] = None
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = ops.Event(), ops.Event()
            e0.record(sp)
            for _ in range(10):
                l.fn(*l.args, sp)
            e1.record(sp); torch.cuda.synchronize()
            ts.append(e0.elapsed_ms(e1) * 100)
        ts.sort()
        out[f"{idx}:{t}"] = round(ts[2], 1)
print(os.environ.get("ICAF_LIB", "default").split("/")[-1], out)
