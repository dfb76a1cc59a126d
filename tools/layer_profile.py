#!/usr/bin/env python3
"""Per-launch HIP-event timing of one forward plan (GPU box): which layers dominate and how far each is from the
MFMA / HBM roofs.  Usage: python tools/layer_profile.py [--model s --batch 32 --size 640 --dtype bf16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
import yaml    # noqa: E402

from icafusion_amd import ops                      # noqa: E402
from icafusion_amd.models.yolo import Model        # noqa: E402
from icafusion_amd.synth import synth_images, synth_state_dict   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s"); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=640); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--sweep", action="store_true", help="time every igemm launch with each tile config")
ap.add_argument("--autotune", action="store_true")
ap.add_argument("--tune-cache", default=None, help="tile choices to load instead of profiles/tune_cache.json (implies --autotune)")
# bench.py's spelling of a workload (tools/gpu_evidence.sh passes the same argument string to both)
ap.add_argument("--height", type=int, default=0); ap.add_argument("--width", type=int, default=0)
ap.add_argument("--loops", type=int, default=1); ap.add_argument("--dataset", default="kaist")
ap.add_argument("--conf", type=float, default=0.0, help="(ignored: no NMS here)")
a = ap.parse_args()
H, W = a.height or a.size, a.width or a.size
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", f"yolov5{a.model}_Transfusion_{a.dataset}.yaml")))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0))
for i in (20, 21, 22):
    m.model[i].crosstransformer[0].loops = a.loops
m = m.to("cuda:0"); m.compute_dtype = dt
m.autotune = a.autotune or bool(a.tune_cache)
if m.autotune:        # the committed igemm configuration choices: the same kernels bench.py launches
    ops.load_tune_cache(a.tune_cache or os.path.join(ROOT, "profiles", "tune_cache.json"))
plan = m.plan_for(a.batch, H, W, "cuda:0")
rgb, ir = synth_images(a.batch, H, W, 0)
plan.inputs[0].copy_(rgb.cuda()); plan.inputs[1].copy_(ir.cuda())
plan.run(); torch.cuda.synchronize()
runs = [[x[1] for x in plan.timed_run()] for _ in range(a.reps)]
acc = [sorted(v)[len(v) // 2] for v in zip(*runs)]       # MEDIAN over the repetitions (a single 0.9 ms hiccup in one of five runs once put 175 us on a 31 us kernel)
rows = []
for l, ms in zip(plan.launches, acc):
    desc = l.name
    if l.fn is ops.lib().icaf_conv2d:
        c = l.keep[0]
        desc = f"{ops.conv_kernel_name(l):26s} {l.name:12s} M={c.B * c.Ho * c.Wo:8d} N={c.Cout:5d} K={c.kh * c.kw * c.Cin:5d} g={c.groups} {c.Ho}x{c.Wo}"
    rows.append((ms, l.flops / (ms * 1e-3) / 1e12 if l.flops else 0.0, l.bytes / (ms * 1e-3) / 1e9, desc))
tot = sum(r[0] for r in rows)
print(f"total kernel time {tot:.3f} ms for batch {a.batch}: {a.batch / tot * 1e3:.0f} pairs/s")
for i, (ms, tf, gb, d) in enumerate(rows):
    print(f"{i:3d} {ms * 1e3:8.1f} us {tf:7.1f} TF {gb:7.0f} GB/s  {d}")

if a.sweep:
    print("\n# tile sweep (us): id1=128x128 id2=128x64 id3=256x32 id4=64x64; '*' = auto choice")
    names = {1: "128x128", 2: "128x64", 3: "256x32", 4: "64x64"}
    sp = ops.current_stream_ptr()
    seen = {}
    for i, l in enumerate(plan.launches):
        if l.fn is not ops.lib().icaf_conv2d:
            continue
        c = l.keep[0]
        key = (c.B * c.Ho * c.Wo, c.Cout, c.kh * c.kw * c.Cin, c.groups, c.sh, c.out_dtype, c.act, c.res is not None)
        if key in seen:
            continue
        seen[key] = i
        auto = "?"
        res = []
        for pipe in (0, 1, 2, 3):
          for t in (1, 2, 3, 4):
            if t == 1 and (c.out_dtype == 0 or c.Cout <= 64):
                continue
            if t == 3 and c.Cout > 32:
                continue
            c.tile = t + 10 * pipe
            l(sp); torch.cuda.synchronize()
            e0, e1 = ops.Event(), ops.Event()
            e0.record(sp)
            for _ in range(10):
                l(sp)
            e1.record(sp)
            res.append((e0.elapsed_ms(e1) * 100, f"p{pipe}:{names[t]}"))
        for big in (25, 26):                               # 8-wavefront tiles
            c.tile = big
            if c.Cout < (128 if big == 25 else 256) or l.fn(*l.args, sp) != 0:
                continue
            torch.cuda.synchronize()
            e0, e1 = ops.Event(), ops.Event()
            e0.record(sp)
            for _ in range(10):
                l(sp)
            e1.record(sp)
            res.append((e0.elapsed_ms(e1) * 100, "p2:256x128" if big == 25 else "p2:256x256"))
        for shape in (1, 2, 3, 4, 5):                      # ctile.hip (3x3 halo-patch kernel); inapplicable -> status != 0
            c.tile = 40 + shape
            if l.fn(*l.args, sp) != 0:
                continue
            torch.cuda.synchronize()
            e0, e1 = ops.Event(), ops.Event()
            e0.record(sp)
            for _ in range(10):
                l(sp)
            e1.record(sp)
            res.append((e0.elapsed_ms(e1) * 100, f"ctile{shape}"))
        c.tile = 0
        best = min(res)
        res.sort()
        print(f"{i:3d} M={key[0]:8d} N={key[1]:5d} K={key[2]:5d} g={key[3]} s={key[4]} best: " +
              " ".join(f"{n}={t:.1f}" for t, n in res[:4]) + " | ctile: " +
              " ".join(f"{n}={t:.1f}" for t, n in res if n.startswith("ctile")))
