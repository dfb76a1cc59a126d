"""Debug: is the output of a batch shard bit-identical to the same rows of the full batch (yolov5l bf16)?  Toggles the
pieces whose arithmetic could depend on the tile the tuner picks."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.models.common import Conv, TransformerFusionBlock
from icafusion_amd.synth import synth_images, synth_state_dict
cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5l_Transfusion_kaist.yaml"))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 3)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
rgb, ir = synth_images(16, 640, 640, seed=3); rgb, ir = rgb.cuda(), ir.cuda()
for tail in (True, False):
    for tune in (False, True):
        ops._TUNE_CACHE.clear()
        m.autotune = tune
        for blk in m.model:
            if isinstance(blk, TransformerFusionBlock):
                blk.fuse_tail = tail
        m.invalidate()
        z = m(rgb, ir)[0].clone()
        zs = m(rgb[8:12].contiguous(), ir[8:12].contiguous())[0]
        d = (z[8:12].float() - zs.float()).abs()
        print(f"fuse_tail={tail} autotune={tune}: shard_equal={torch.equal(z[8:12], zs)} maxdiff={d.max().item():.3e} n_diff={(d > 0).sum().item()}", flush=True)
        if not torch.equal(z[8:12], zs):
            pa, pb = m.plan_for(16, 640, 640), m.plan_for(4, 640, 640)
            for la, lb in zip(pa.launches, pb.launches):
                if la.fn is ops.lib().icaf_conv2d:
                    na, nb = ops.conv_kernel_name(la), ops.conv_kernel_name(lb)
                    if na != nb:
                        print("   ", la.name, na, "vs", nb)
