"""A/B: bottleneck + cv3 as one persistent kernel (Conv.fuse_stem2) vs stem -> conv+chained 1x1 (yolov5s bf16, batch 32, 640x640)."""
import os
import sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch   # noqa: E402
import yaml    # noqa: E402
from icafusion_amd import ops                      # noqa: E402
from icafusion_amd.models.yolo import Model        # noqa: E402
from icafusion_amd.models.common import Conv, C3      # noqa: E402
from icafusion_amd.synth import synth_state_dict   # noqa: E402

cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5s_Transfusion_kaist.yaml"))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
m.autotune = True; m.use_graph = True
ops.load_tune_cache(f"{R}/profiles/tune_cache.json")
st = torch.cuda.Stream(); sp = st.cuda_stream
for rnd in range(2):
    for on in (True, False):
        C3.fuse_cv3 = on
        m.invalidate()
        plan = m.plan_for(32, 640, 640, "cuda:0")
        for _ in range(3):
            plan.run(sp)
        e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
        for _ in range(20):
            plan.run(sp)
        e1.record(sp)
        torch.cuda.synchronize()
        fwd = e0.elapsed_ms(e1) / 20
        ls = plan.launches[1:2] if on else plan.launches[1:3]         # bottleneck[+cv3] | bottleneck, cv3
        for l in ls:
            l(sp)
        torch.cuda.synchronize()
        e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
        for _ in range(10):
            for l in ls:
                l(sp)
        e1.record(sp); torch.cuda.synchronize()
        print(f"fuse_cv3={on}: forward {fwd:.3f} ms, bottleneck + cv3 {e0.elapsed_ms(e1) * 100:.1f} us, {len(plan.launches)} launches", flush=True)
