"""The serving loop on the MI355X: DetectionPipeline (forward on one stream, NMS [+ detections all-gather] of the previous batch
on a second one), the RCCL collective exercised with a one-rank process group, and the byte-capped LRU of execution plans."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import REPO, load_cfg                               # noqa: E402
from icafusion_amd.models.yolo import Model                      # noqa: E402
from icafusion_amd.pipeline import DetectionPipeline             # noqa: E402
from icafusion_amd.synth import synth_images, synth_state_dict   # noqa: E402
from icafusion_amd.utils.general import non_max_suppression      # noqa: E402

DEV = "cuda:0"


def build(yaml_name, dtype, seed=0):
    m = Model(load_cfg(yaml_name)).eval()
    m.load_state_dict(synth_state_dict(m, seed))
    m = m.to(DEV)
    m.compute_dtype = dtype
    return m


@pytest.mark.parametrize("overlap", [True, False])
def test_pipeline_steps_equal_sequential_forward_plus_nms(overlap):
    """Five different batches through the overlapped pipeline: every step's detections equal forward -> NMS run back to back,
    and the tensors a step returned are still intact after the NEXT step was enqueued (each in-flight slot owns its buffers)."""
    m = build("yolov5s_Transfusion_FLIR.yaml", torch.bfloat16)
    m.use_graph = True
    B, H, W = 4, 320, 384
    pipe = DetectionPipeline(m, B, H, W, DEV, conf_thres=0.25, iou_thres=0.45, overlap=overlap)
    batches = [synth_images(B, H, W, seed=40 + k) for k in range(5)]
    got = []
    for rgb, ir in batches:
        pipe.inputs[0].copy_(rgb.to(DEV)); pipe.inputs[1].copy_(ir.to(DEV))
        torch.cuda.current_stream().synchronize()
        got.append(tuple(t[0] for t in pipe.step()))                # (world = 1: drop the rank axis)
        if len(got) >= 2 and overlap:                        # step n-1's tensors must survive step n being enqueued
            pipe.synchronize()
            det, count = got[-2]
            prev = [det[k, :n].clone() for k, n in enumerate(count.tolist())]
            rgb0, ir0 = batches[len(got) - 2]
            m.static_outputs = False
            want = non_max_suppression(m(rgb0.to(DEV), ir0.to(DEV))[0], 0.25, 0.45)
            m.static_outputs = True
            assert all(torch.equal(a, b) for a, b in zip(prev, want))
        if not overlap:
            pipe.synchronize()
            got[-1] = tuple(t.clone() for t in got[-1])
    pipe.synchronize()
    m.static_outputs = False
    rgb, ir = batches[-1]
    want = non_max_suppression(m(rgb.to(DEV), ir.to(DEV))[0], 0.25, 0.45)
    det, count = got[-1]
    assert sum(count.tolist()) > 0
    assert all(torch.equal(det[k, :n], w) for (k, n), w in zip(enumerate(count.tolist()), want))


@pytest.mark.parametrize("depth", [2, 3])
def test_pipeline_with_several_batches_in_flight(depth):
    """depth plans / forward streams: six different batches, each step's detections equal forward -> NMS run one at a time, and a
    step's tensors stay intact until its slot is reused `depth` steps later."""
    m = build("yolov5s_Transfusion_FLIR.yaml", torch.bfloat16)
    m.use_graph = True
    B, H, W = 4, 320, 352
    pipe = DetectionPipeline(m, B, H, W, DEV, conf_thres=0.25, iou_thres=0.45, depth=depth)
    assert len({p.outputs[0].data_ptr() for p in pipe.plans}) == depth            # separate buffers per batch in flight
    batches = [synth_images(B, H, W, seed=60 + k) for k in range(6)]
    outs = []
    for rgb, ir in batches:
        outs.append(tuple(t[0] for t in pipe.submit(rgb.to(DEV), ir.to(DEV))))    # copies on the slot's forward stream, no host sync; rank axis dropped
    pipe.synchronize()
    m.static_outputs = False
    ref = build("yolov5s_Transfusion_FLIR.yaml", torch.bfloat16)                  # an independent model instance, one batch at a time
    for k in range(len(batches) - depth, len(batches)):                           # the last `depth` steps still own their buffers
        rgb, ir = batches[k]
        want = non_max_suppression(ref(rgb.to(DEV), ir.to(DEV))[0], 0.25, 0.45)
        det, count = outs[k]
        assert sum(count.tolist()) > 0
        assert all(torch.equal(det[i, :n], w) for (i, n), w in zip(enumerate(count.tolist()), want)), f"step {k}"


@pytest.mark.parametrize("depth", [1, 2])
def test_pipeline_fed_with_pinned_uint8_host_batches(depth):
    """DetectionPipeline(u8=True).submit_u8: pinned host uint8 (B, 6, H, W) batches through the copy stream STRAIGHT into the input buffer of
    one of depth * (1 + EXTRA_PLANS) plans — one no forward in flight is reading: one PCIe copy per batch, no device-to-device hop (reference
    test.py:116-123 copies, casts, divides and splits every batch).  Seven different batches, none of them waited for before the next is
    submitted: every step's detections equal Model.forward_u8 + NMS of that batch run on their own."""
    m = build("yolov5s_Transfusion_FLIR.yaml", torch.bfloat16)
    m.use_graph = True
    B, H, W = 4, 320, 352
    pipe = DetectionPipeline(m, B, H, W, DEV, conf_thres=0.25, iou_thres=0.45, depth=depth, u8=True)
    assert pipe.nplans > depth and len({p.inputs[0].data_ptr() for p in pipe.plans}) == pipe.nplans and not hasattr(pipe, "stage")
    g = torch.Generator().manual_seed(5)
    host = [torch.randint(0, 256, (B, 6, H, W), dtype=torch.uint8, generator=g).pin_memory() for _ in range(7)]
    outs = [tuple(t[0] for t in pipe.submit_u8(h)) for h in host]
    pipe.synchronize()
    ref = build("yolov5s_Transfusion_FLIR.yaml", torch.bfloat16)
    for k in range(len(host) - pipe.nplans, len(host)):                           # the last `nplans` steps still own their output buffers
        want = non_max_suppression(ref.forward_u8(host[k].to(DEV))[0], 0.25, 0.45)
        det, count = outs[k]
        assert sum(count.tolist()) > 0
        assert all(torch.equal(det[i, :n], w) for (i, n), w in zip(enumerate(count.tolist()), want)), f"step {k}"


_RCCL_SCRIPT = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, {repo!r})
import torch.distributed as dist
from icafusion_amd import dist as D
from icafusion_amd.models.yolo import Model
from icafusion_amd.pipeline import DetectionPipeline
from icafusion_amd.synth import synth_images, synth_state_dict
import yaml
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
cfg = yaml.safe_load(open(os.path.join({repo!r}, "models", "transformer", "yolov5s_Transfusion_kaist.yaml")))
m = Model(cfg).eval(); m.load_state_dict(synth_state_dict(m, 0)); m = m.to("cuda:0"); m.compute_dtype = torch.bfloat16; m.use_graph = True
B, H, W = 4, 320, 320
pipe = DetectionPipeline(m, B, H, W, "cuda:0", conf_thres=0.1, iou_thres=0.5, world=1, overlap=True, force_gather=True, depth={depth})
runners = pipe.deep_runners if pipe.depth > 1 else pipe.runners            # the slot step k used
blk = B * 300 * 6 + B
assert pipe.group == (pipe.depth if pipe.depth > 1 else 1) and pipe.gather_stream is not pipe.nms_stream
outs, clones = [], []
def feed(k):
    rgb, ir = synth_images(B, H, W, seed=70 + k)
    pipe.inputs[0].copy_(rgb.cuda()); pipe.inputs[1].copy_(ir.cuda()); torch.cuda.current_stream().synchronize()
# pass 1: synchronised after every step (a group cut short is sent by synchronize())
for k in range(4):
    feed(k)
    gen = pipe.gen
    outs.append(pipe.step())
    pipe.synchronize()
    det_all, count_all = outs[-1]
    r = runners[k % len(runners)]
    assert det_all.shape == (1, B, 300, 6) and count_all.shape == (1, B) and count_all.dtype == torch.int32
    assert torch.equal(det_all[0], r.det) and torch.equal(count_all[0], r.count), "all-gather of one rank must return that rank's block"
    assert r.det.data_ptr() == r.block.data_ptr()                              # NMS wrote into the block that travelled: no packing step
    slot = k % pipe.group
    assert r.block.data_ptr() == pipe.group_block.data_ptr() + 4 * blk * (k % len(runners))     # the slots' blocks lie side by side
    assert int(r.count.sum()) > 0
    assert det_all.data_ptr() == pipe.gathered[gen].data_ptr() + 4 * blk * slot   # the collective wrote the pipeline's gathered buffer
    clones.append((det_all.clone(), count_all.clone()))
g0 = pipe.gathers
assert g0 == 4
# pass 2: the same four batches with no synchronisation in between: ONE collective per group of `group` steps, on the gather stream
outs2 = []
for k in range(4):
    feed(k)
    outs2.append(pipe.step())
pipe.synchronize()
assert pipe.gathers - g0 == 4 // pipe.group
for k in range(4):
    if k >= 4 - 2 * pipe.group:                     # (two generations of the gathered buffer: the last two groups are still intact)
        assert torch.equal(outs2[k][0], clones[k][0]) and torch.equal(outs2[k][1], clones[k][1]), k
dist.barrier(); dist.destroy_process_group()
print("RCCL_WORLD1_OK", [int(c.sum()) for _, c in outs])
'''


@pytest.mark.parametrize("depth", [1, 2])
def test_rccl_all_gather_runs_on_its_own_stream_with_one_rank(depth):
    """`nccl` (= RCCL) process group of world size 1: DetectionPipeline(force_gather=True) sends the detection blocks through
    dist.all_gather_into_tensor — ONE collective per group of `depth` steps on the gather stream, behind the group's last NMS; the
    collective, its stream ordering against the NMS kernels and the two generations of gathered buffers run on hardware although no
    second GPU exists (a subprocess: it owns the process group).  depth = 2 is what `bench.py --gpus N` runs."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(repo=REPO, depth=depth)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_plan_cache_does_not_grow_over_many_shapes():
    """A rectangular-batch validation run of yolov5l meets a new (B, H, W) every few batches; with the LRU cap the device memory
    held by plans stays bounded (one never-freed plan per shape, ~40 GB each at 1280x1280 b16)."""
    m = build("yolov5l_Transfusion_kaist.yaml", torch.bfloat16)
    shapes = [(4, 544, 672), (4, 512, 640), (4, 640, 640), (4, 640, 512), (3, 544, 672), (4, 576, 640), (4, 608, 672), (2, 640, 640)]
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()                  # model parameters
    first = m.plan_for(*shapes[0], device=DEV)
    one = first.nbytes
    del first
    m.plan_cache_bytes = int(2.6 * one)
    peak = []
    for s in shapes * 2:
        rgb, ir = synth_images(s[0], s[1], s[2], seed=1)
        z = m(rgb.to(DEV), ir.to(DEV))[0]
        assert torch.isfinite(z).all()
        held = sum(p.nbytes for p in m._plans.values())
        assert held <= m.plan_cache_bytes and len(m._plans) <= 3
        del z, rgb, ir
        torch.cuda.synchronize()
        peak.append(torch.cuda.memory_allocated() - base)
    # live device memory = the cached plans (<= cap) + packed weights (one set per dtype, ~ the parameters again in bf16) + NMS-free
    # outputs; it must not creep upwards from pass to pass: without the cap the 16 forwards would pin 8 plans (~8 x one)
    packed = base
    assert max(peak) <= m.plan_cache_bytes + packed + 0.25 * one, (max(peak), m.plan_cache_bytes, packed, one)
    assert max(peak[8:]) <= m.plan_cache_bytes + packed + 0.25 * one
    assert max(peak) < 5 * one
