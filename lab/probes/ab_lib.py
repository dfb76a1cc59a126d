"""Forward time + per-launch times of the bench plan (yolov5s bf16, batch 32, 640x640, committed tune cache) as JSON on stdout.
Run it under different ICAF_LIB settings on ONE box to A/B kernel variants:  tools/build_variant.py, lab/probes/ab_diff.py."""
import json
import os
import sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch   # noqa: E402
import yaml    # noqa: E402
from icafusion_amd import ops                      # noqa: E402
from icafusion_amd.models.yolo import Model        # noqa: E402
from icafusion_amd.synth import synth_state_dict   # noqa: E402

cfg = yaml.safe_load(open(f"{R}/models/transformer/yolov5s_Transfusion_kaist.yaml"))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16
m.autotune = True; m.use_graph = True
# ICAF_AB_TUNE: unset = the committed tile choices; "fresh" = tune every layer on the spot (new candidates get their chance);
# a path = load it if it exists, tune the rest, save it there
tune = os.environ.get("ICAF_AB_TUNE", "")
if not tune:
    ops.load_tune_cache(f"{R}/profiles/tune_cache.json")
elif tune != "fresh" and os.path.exists(tune):
    ops.load_tune_cache(tune)
plan = m.plan_for(32, 640, 640, "cuda:0")
if tune and tune != "fresh":
    ops.save_tune_cache(tune)
st = torch.cuda.Stream(); sp = st.cuda_stream
fw = []
for rep in range(3):
    for _ in range(3):
        plan.run(sp)
    e0, e1 = ops.Event(), ops.Event(); e0.record(sp)
    for _ in range(20):
        plan.run(sp)
    e1.record(sp); torch.cuda.synchronize()
    fw.append(e0.elapsed_ms(e1) / 20)
acc = None
for _ in range(5):
    r = plan.timed_run()
    acc = [x[1] for x in r] if acc is None else [p + x[1] for p, x in zip(acc, r)]
rows = []
for l, ms in zip(plan.launches, acc):
    name = l.name + (" " + ops.conv_kernel_name(l) if l.fn is ops.lib().icaf_conv2d else "")
    rows.append((name, ms / 5 * 1e3))
print(json.dumps({"lib": os.environ.get("ICAF_LIB", "default"), "forward_ms": fw, "launches": rows}))
