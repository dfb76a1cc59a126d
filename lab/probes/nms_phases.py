#!/usr/bin/env python3
"""Phase clocks of nms_walk_kernel (image 0) on the bench workload's predictions: build nms.hip with -DICAF_NMS_DEBUG into
libicaf_nmsdbg.so and run  ICAF_LIB=icafusion_amd/lib/libicaf_nmsdbg.so python lab/probes/nms_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.synth import synth_images, synth_state_dict
from icafusion_amd.utils.general import nms_device
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", "yolov5s_Transfusion_kaist.yaml")))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
rgb, ir = synth_images(32, 640, 640, seed=100)
z = m(rgb.cuda(), ir.cuda())[0]
for conf in (0.1, 0.001):
    det, count, keep = nms_device(z, conf, 0.5)
    torch.cuda.synchronize()
    print("conf", conf, "counts", count[:4].tolist())
