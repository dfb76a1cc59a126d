#!/usr/bin/env python3
"""Refinement of the launch-configuration choices against the clock `bench.py` reports as `value`: forward throughput with TWO batches in
flight (two plans, two hipGraphs, two streams).  The per-launch tuner (ops.autotune_conv) times a launch alone and tools/graph_tune.py one
forward alone; with a second forward sharing the chip the ranking changes at the margin (a configuration with fewer, larger workgroups wins
alone and starves the co-running forward: docs/HISTORY.md section 15).  Starting from a committed cache, for the layer
signatures that cost most, both plans are re-captured with every alternative configuration and a change is kept only if the two-in-flight rate
improves by more than --eps, twice in a row.  All configurations of a layer produce the same bits, so this is a pure scheduling choice.

    python tools/throughput_tune.py [--model s --batch 32 --size 640 --dtype bf16 --top 20] [--seed-cache profiles/tune_cache.json] --out gpurun_out/tune_tp.json
"""
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
ap.add_argument("--height", type=int, default=0); ap.add_argument("--width", type=int, default=0)
ap.add_argument("--loops", type=int, default=1); ap.add_argument("--dataset", default="kaist")
ap.add_argument("--top", type=int, default=20, help="number of layer signatures (by time) to refine")
ap.add_argument("--eps", type=float, default=0.004, help="relative improvement of the two-in-flight rate required to accept a change")
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--seed-cache", default=None)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tune_tp.json"))
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", f"yolov5{a.model}_Transfusion_{a.dataset}.yaml")))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0))
for i in (20, 21, 22):
    m.model[i].crosstransformer[0].loops = a.loops
m = m.to("cuda:0"); m.compute_dtype = dt
H, W = a.height or a.size, a.width or a.size
seed = a.seed_cache or os.path.join(ROOT, "profiles", "tune_cache.json")
if os.path.exists(seed):
    ops.load_tune_cache(seed)
m.autotune = True
m.use_graph = False                               # (captured here, after every change)
plans = [m.plan_for(a.batch, H, W, "cuda:0", slot=s) for s in (0, 1)]
rgb, ir = synth_images(a.batch, H, W, 0)
for p in plans:
    p.inputs[0].copy_(rgb.cuda()); p.inputs[1].copy_(ir.cuda())
streams = [torch.cuda.Stream() for _ in plans]


def rate(reps=3):
    """ms per batch with the two plans alternating on their streams (best of `reps` bursts of --steps batches)"""
    for p in plans:
        p.capture()
    for k in range(4):
        plans[k & 1].run(streams[k & 1].cuda_stream)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        e0.record(torch.cuda.current_stream())
        for s in streams:
            s.wait_event(e0)
        for k in range(a.steps):
            plans[k & 1].run(streams[k & 1].cuda_stream)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / a.steps)
    return best


fn = ops.lib().icaf_conv2d
groups = {}
for pi, p in enumerate(plans):
    for l, (name, ms, fl, nb) in zip(p.launches, p.timed_run()):
        if l.fn is fn:
            g = groups.setdefault(ops._conv_signature(l.keep[0]), {"launches": [], "ms": 0.0})
            g["launches"].append(l)
            if pi == 0:
                g["ms"] += ms
base = rate()
print(f"start: {base:.4f} ms per batch with two in flight ({a.batch / base * 1e3:.0f} pairs/s forward only)")
for sig, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:a.top]:
    args0 = g["launches"][0].keep[0]
    cur = args0.tile
    sp = streams[0].cuda_stream
    base = rate(reps=2)                           # re-measured per group, right before its candidates: the chip's clock drifts over a long run
    best_c, best_t = cur, base
    for c in [c for c in ops.conv_candidates(args0) if c != cur]:
        for l in g["launches"]:
            l.keep[0].tile = c
        if g["launches"][0].fn(*g["launches"][0].args, sp) != 0:       # configuration not applicable to this layer
            continue
        torch.cuda.synchronize()
        t = rate(reps=2)
        if t < best_t:
            best_c, best_t = c, t
    accept = False
    if best_c != cur and best_t < base * (1 - a.eps):
        for l in g["launches"]:
            l.keep[0].tile = best_c
        t2 = rate()                                                   # confirm: the rate moves by a few tenths of a per cent from burst to burst
        for l in g["launches"]:
            l.keep[0].tile = cur
        b2 = rate()
        accept = t2 < b2 * (1 - a.eps)
        if accept:
            best_t = t2
    for l in g["launches"]:
        l.keep[0].tile = best_c if accept else cur
    if accept:
        ops._TUNE_CACHE[sig] = best_c
        print(f"  M={sig[0]:8d} N={sig[1]:4d} Cin={sig[2]:4d} k={sig[3]} x{len(g['launches']) // 2}: {cur} -> {best_c}   {b2:.4f} -> {t2:.4f} ms (interleaved)")
final = rate()
print(f"final: {final:.4f} ms per batch with two in flight ({a.batch / final * 1e3:.0f} pairs/s forward only)")
os.makedirs(os.path.dirname(a.out), exist_ok=True)
ops.save_tune_cache(a.out)
print("saved", a.out)
