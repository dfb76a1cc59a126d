"""A/B one class-level plan flag on the bench plan (yolov5s bf16, batch 32, 640x640, committed tune cache + in-situ tuning of
new signatures):  python lab/probes/ab_flag.py Bottleneck.fuse_widths "(32, 64)" "(32,)" """
import os
import sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch   # noqa: E402
import yaml    # noqa: E402
from icafusion_amd import ops                      # noqa: E402
from icafusion_amd.models.yolo import Model        # noqa: E402
from icafusion_amd.models import common            # noqa: E402
from icafusion_amd.synth import synth_state_dict   # noqa: E402

cls, attr = sys.argv[1].split(".")
values = [eval(v) for v in sys.argv[2:]]
cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5s_Transfusion_kaist.yaml"))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
m.autotune = True; m.use_graph = True
ops.load_tune_cache(f"{R}/profiles/tune_cache.json")
st = torch.cuda.Stream(); sp = st.cuda_stream
for rnd in range(2):
    for v in values:
        setattr(m if cls == "Model" else getattr(common, cls), attr, v)      # (Model.x: an attribute of the model instance)
        m.invalidate()
        plan = m.plan_for(32, 640, 640, "cuda:0")
        for _ in range(3):
            plan.run(sp)
        e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
        for _ in range(20):
            plan.run(sp)
        e1.record(sp); torch.cuda.synchronize()
        print(f"{sys.argv[1]}={v}: forward {e0.elapsed_ms(e1) / 20:.3f} ms, {len(plan.launches)} launches", flush=True)
