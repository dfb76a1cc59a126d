#!/usr/bin/env python3
"""Launch ONE conv layer of the yolov5s plan a few times (for rocprofv3 --pmc runs).  args: launch index, tile id, reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.synth import synth_state_dict
idx, tile, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", "yolov5s_Transfusion_kaist.yaml")))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
plan = m.plan_for(32, 640, 640, "cuda:0")
l = plan.launches[idx]
l.keep[0].tile = tile
sp = ops.current_stream_ptr()
for _ in range(reps):
    l(sp)
torch.cuda.synchronize()
print(ops.conv_kernel_name(l), l.name)
