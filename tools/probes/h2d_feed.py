"""The host-fed serving loop (bench.py's h2d_feed) under several feeds, one process, one box:
    python tools/probes/h2d_feed.py [--feeds 0,8,16,32,64] [bench.py arguments]
feed 0 = the DMA engine (Tensor.copy_ on the copy stream), n > 0 = icaf_feed_copy with n resident workgroups.  Prints one JSON line per feed
and the no-feed rate of the same pipeline shape (inputs resident) before and after."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
feeds = [0, 8, 16, 32, 64]
argv = sys.argv[1:]
if "--feeds" in argv:
    i = argv.index("--feeds")
    feeds = [int(x) for x in argv[i + 1].split(",") if x not in ("", "none")]
    del argv[i:i + 2]
burn = 0
if "--burn" in argv:                      # take that many streams from torch's pool first: shifts every later stream's place in the pool
    i = argv.index("--burn")
    burn = int(argv[i + 1])
    del argv[i:i + 2]
skip_resident = "--skip-resident" in argv
if skip_resident:
    argv.remove("--skip-resident")
eager = "--eager" in argv                 # no hipGraph: every kernel launched from the host
if eager:
    argv.remove("--eager")
no_branch = "--no-branch" in argv         # hipGraph without the parallel DMFF / Detect branches
if no_branch:
    argv.remove("--no-branch")
sys.argv = [sys.argv[0]] + argv
import torch  # noqa: E402
import yaml  # noqa: E402
import bench  # noqa: E402
from icafusion_amd import ops, pipeline as P  # noqa: E402
from icafusion_amd.models.yolo import Model  # noqa: E402
from icafusion_amd.synth import synth_images, synth_state_dict  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", f"yolov5{args.model}_Transfusion_{args.dataset}.yaml")))
model = Model(cfg).eval()
model.load_state_dict(synth_state_dict(model, seed=0))
for i in (20, 21, 22):
    model.model[i].crosstransformer[0].loops = args.loops
model = model.to(dev)
model.compute_dtype = bench.DT[args.dtype]
model.static_outputs = True
model.autotune = True
model.use_graph = not eager
model.branch_dmff = not no_branch
cache = args.tune_cache or os.path.join(ROOT, "profiles", "tune_cache.json")
if os.path.exists(cache):
    ops.load_tune_cache(cache)
B, H, W = args.batch, args.height, args.width


def resident():
    pipe = P.DetectionPipeline(model, B, H, W, dev, conf_thres=args.conf, iou_thres=args.iou, world=1, depth=args.depth)
    rgb, ir = synth_images(B, H, W, seed=100)
    for pl in pipe.plans:
        pl.inputs[0].copy_(rgb.to(dev))
        pl.inputs[1].copy_(ir.to(dev))
    for _ in range(args.warmup):
        pipe.step()
    rates = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.step()
        pipe.synchronize()
        torch.cuda.synchronize()
        rates.append(B * args.steps / (time.perf_counter() - t0))
    return round(sorted(rates)[1], 1)


burned = [torch.cuda.Stream(device=dev) for _ in range(burn)]
if not skip_resident:
    print(json.dumps({"resident_pairs_per_s": resident()}), flush=True)
for f in feeds:
    P.FEED_WGS = f
    r = bench.h2d_feed(model, args, B, H, W, dev)
    r.pop("note", None)
    print(json.dumps({"feed_wgs": f, "graph": not eager, "branches": not no_branch, **r}), flush=True)
if not skip_resident:
    print(json.dumps({"resident_pairs_per_s": resident()}), flush=True)
