import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, yaml
from icafusion_amd import ops
from icafusion_amd.models.yolo import Model
from icafusion_amd.models.common import Conv, Bottleneck
from icafusion_amd.synth import synth_images, synth_state_dict
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for mdl in ("l", "m"):
    cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5{mdl}_Transfusion_kaist.yaml"))
    m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
    m.autotune = True; m.use_graph = True
    for chain in (True, False):
        Conv.chain_fuse = chain
        m.invalidate()
        plan = m.plan_for(32, 640, 640, "cuda:0")
        st = torch.cuda.Stream(); sp = st.cuda_stream
        for _ in range(3): plan.run(sp)
        e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
        for _ in range(10): plan.run(sp)
        e1.record(sp); print(mdl, "chain", chain, "forward ms", round(e0.elapsed_ms(e1) / 10, 3), "launches", len(plan.launches))
