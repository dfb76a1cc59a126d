#!/usr/bin/env python3
"""bench.py — RGB/IR image-pairs/sec of the ICAFusion hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One step = one pass of the hot path over one batch of synthetic pairs already resident in HBM:
two-stream backbone -> 3x DMFF -> PANet head -> Detect (one hipGraph replay) -> device NMS -> [N>1: RCCL all-gather
of the detection blocks].  Steps are pipelined as a serving loop pipelines them: `--depth` batches in flight (default 2, each with
its own plan / buffers / forward stream, so the low-occupancy tail of one forward overlaps the full-width layers of the next) and
the NMS of a batch on a further stream; every step does all of its work inside the timed region.  Default workload = BASELINE.json configs[1]: yolov5s + DMFF, bf16, batch 32, 640x640.
Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the CPU oracle = a port of the reference's algorithm, timed on this host's cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml   # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
# algorithmic FLOPs per pair (2*MAC of conv + linear + bmm), SURVEY.md §8d / BASELINE.md
GFLOP_PER_PAIR = {("s", 640, 640): 30.08, ("s", 512, 640): 24.61, ("l", 640, 640): 192.2, ("l", 1280, 1280): 738.9}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: this many pairs per step over all GPUs, split contiguously "
                    "(dist.shard_range: BASELINE config 3 = 256 -> 32/GPU on 8, config 5 = 128 -> 16/GPU); overrides --batch")
    ap.add_argument("--force-gather", action="store_true", help="one rank only: initialise a 1-rank RCCL group and send every step's detection "
                    "block through the all-gather anyway (what a rank of an N > 1 run pays for the collective and its stream, measured on one GPU)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive serving measurement (pinned host uint8 batches -> device, "
                    "overlapped with the forwards in flight; reported as `h2d_feed`, never as `value`)")
    ap.add_argument("--height", type=int, default=640)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--model", default="s", choices=["n", "s", "m", "l"])
    ap.add_argument("--dataset", default="kaist")
    ap.add_argument("--dtype", default="bf16", choices=list(DT))
    ap.add_argument("--loops", type=int, default=1)
    ap.add_argument("--conf", type=float, default=0.1, help="detect_twostream.py default")
    ap.add_argument("--iou", type=float, default=0.5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--repeats", type=int, default=9, help="the timed region (K steps between barriers) is run at least this many times; "
                    "`value` is the median run, min / max are reported beside it")
    ap.add_argument("--min-timed-seconds", type=float, default=3.0, help="keep repeating the timed region (same K steps each) until this much "
                    "timed GPU work has been done: one region is K x ~2 ms, too short for a utilisation sampler to see (0 = exactly --repeats regions)")
    ap.add_argument("--no-fuse-tail", action="store_true", help="materialise DMFF's merged tensor instead of the fused-tail GEMM")
    ap.add_argument("--no-overlap", action="store_true", help="run NMS on the forward stream (no cross-batch overlap)")
    ap.add_argument("--depth", type=int, default=2, help="batches in flight, each with its own plan (buffers, hipGraph) and forward stream: the "
                    "low-occupancy tail of one forward (20x20 layers, DMFF, Detect) overlaps the full-width layers of the next; 1 = one batch at a time")
    ap.add_argument("--fold-upsample", action="store_true", help="head rows Upsample -> Concat -> C3: run the up-sampled half of the 1x1 at low resolution")
    ap.add_argument("--tune-cache", default=None, help="json file: load igemm tile choices if present, save after tuning")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement (latency_ms_b1)")
    ap.add_argument("--dry-run", action="store_true", help="launcher check: start the ranks, initialise the process group (RCCL with GPUs, gloo "
                    "without), all-gather one detection block, print {n_gpus, backend} and exit - no model, no timing")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks ourselves (one process per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1) and pass rank 0's JSON line through.  The driver's own
    `python -m torch.distributed.run ... bench.py --gpus N` form sets WORLD_SIZE and never comes here."""
    import subprocess
    # --standalone: torchrun starts its own c10d rendezvous on a port IT picks and keeps (no bind-close-reuse race on a busy host),
    # on the loopback address (the container's hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # (this image's host driver only supports dmabuf IPC: without the variable RCCL fails with hipIpcGetMemHandle: invalid argument.
    #  It is exported on the GPU boxes already; a value the user set is never overridden.)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"[bench] --gpus {args.gpus} without a torchrun environment: launching {args.gpus} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Rank start-up, rank -> GPU binding, process group and the one collective of the path, without the model."""
    from icafusion_amd import dist as D
    import torch.distributed as tdist
    rank, world, local = D.init_from_env()
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(local)
    max_det = 300
    G = args.global_batch
    B = D.padded_local_batch(G, world) if G else 4
    block, det, count = D.detection_block(B, max_det, dev)
    det.fill_(float(rank + 1))
    count.fill_(rank + 1)
    if G:                              # strong scaling: every pair carries its GLOBAL index; ranks short of B pairs leave padding rows behind
        lo, hi = D.shard_range(G, rank, world)
        for k in range(hi - lo):
            det[k].fill_(float(lo + k))
            count[k] = lo + k + 1
    det_all, count_all = D.gather_detections(det, count, block=block)
    ok = det_all.shape[0] == world and all(int(count_all[r, 0]) == (r + 1 if not G else D.shard_range(G, r, world)[0] + 1) for r in range(world))
    if G:                              # the global view: exactly G pairs, in global order, padding rows dropped
        det_g, count_g = D.gathered_to_global(det_all, count_all, G)
        ok = ok and det_g.shape[0] == G and all(float(det_g[i, 0, 0]) == i and int(count_g[i]) == i + 1 for i in range(G))
    if world > 1:
        tdist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "requested_gpus": args.gpus, "gather_ok": bool(ok), "global_batch": G or None,
                          "local_batches": [D.shard_range(G, r, world)[1] - D.shard_range(G, r, world)[0] for r in range(world)] if G else None,
                          "backend": tdist.get_backend() if world > 1 else None, "device": str(dev)}))
    if world > 1:
        tdist.destroy_process_group()
    return 0 if ok and world == args.gpus else 1


def sq_counters(workload):
    """The committed SQ-counter summary of this workload (tools/gpu_pmc_sq.sh: two rocprofv3 --pmc passes of the same command line, merged;
    tools/gpu_evidence.sh takes them in the same call as the bench line and installs them as profiles/pmc_sq*.json first): per kernel name
    MFMA-pipe busy, VALU issue fraction, wave-parked fraction.  PMC counters cannot be read from inside this process."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_sq*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("kernels"):
            return d["kernels"], os.path.basename(f)
    return {}, None


def binding_roof(fracs):
    """name of the largest of the roofs a kernel (or the forward) was measured against: what it is closest to being bound by"""
    best = max(((v, k) for k, v in fracs.items() if v is not None), default=(None, None))
    return best[1]


def cpu_baseline(cfg, sd, args, loops):
    """Oracle forward + oracle NMS on host cores, bounded sample (checker code used as the measured CPU port).
    `sd` is the FUSED state_dict (BatchNorm folded into the convs): the reference serves `.fuse().eval()` models
    (models/experimental.py:119), i.e. one conv + bias op per layer, not conv + a separate BN pass.
    The thread count is calibrated first: with every hardware thread of a large host, torch's CPU convs oversubscribe
    badly on these small layers, so the best of a few counts is used and reported as `cores`."""
    from icafusion_amd.synth import synth_images
    from oracle import icaf_oracle as oracle
    om = oracle.OracleModel(cfg, sd, loops=loops)
    r1, i1 = synth_images(1, args.height, args.width, seed=0)
    ncpu = os.cpu_count() or 1
    best_thr, best_t = 1, float("inf")
    for thr in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(thr)
        om.forward(r1, i1)
        t0 = time.perf_counter()
        om.forward(r1, i1)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_thr, best_t = thr, dt
        if dt > 5.0:
            break
    torch.set_num_threads(best_thr)
    bs = 4 if best_t < 1.0 else 1       # (yolov5l at 1280 x 1280 takes seconds per pair: the bounded sample is then single pairs)
    rgb, ir = synth_images(bs, args.height, args.width, seed=0)
    t_fwd = t_nms = 0.0
    n = 0
    t_start = time.perf_counter()
    while n < 16 and (n == 0 or (time.perf_counter() - t_start) < args.cpu_seconds):
        t0 = time.perf_counter()
        z = om.forward(rgb, ir)[0]
        t1 = time.perf_counter()
        oracle.non_max_suppression(z.numpy(), args.conf, args.iou)
        t2 = time.perf_counter()
        t_fwd += t1 - t0
        t_nms += t2 - t1
        n += 1
    pairs = n * bs
    # BASELINE configs[0]: ONE pair through forward + NMS (detect_twostream.py's frame loop), the CPU twin of `latency_b1`
    lat = []
    t_start = time.perf_counter()
    while len(lat) < 5 and (not lat or (time.perf_counter() - t_start) < max(2.0, args.cpu_seconds / 4)):
        t0 = time.perf_counter()
        z1 = om.forward(r1, i1)[0]
        oracle.non_max_suppression(z1.numpy(), args.conf, args.iou)
        lat.append(time.perf_counter() - t0)
    lat.sort()
    b1 = lat[len(lat) // 2]
    return {"value": round(pairs / (t_fwd + t_nms), 3), "unit": "pairs/s", "cores": best_thr, "host_cpus": ncpu,
            "kind": "port", "forward_pairs_per_s": round(pairs / t_fwd, 3), "nms_ms_per_pair": round(1e3 * t_nms / pairs, 3),
            "single_pair": {"pairs_per_s": round(1.0 / b1, 3), "latency_ms": round(1e3 * b1, 2), "samples": len(lat),
                            "note": "BASELINE configs[0]: one pair, forward + NMS, same threads (the CPU twin of latency_b1)"},
            "sample": f"{n} batches of {bs} pairs, {args.height}x{args.width}, fp32 torch-CPU oracle forward + C/numpy NMS "
                      f"(oracle/icaf_oracle.py) on the BN-folded weights (= the reference's .fuse().eval() work), same synthetic weights/inputs recipe, "
                      f"{best_thr} threads (best of 8/16/32/64)"}


def h2d_feed(model, args, B, H, W, dev):
    """The serving loop fed from HOST memory, as the reference's loops are (detect_twostream.py:70-80, test.py:116-123 copy every batch
    to the device, then forward, then NMS): pinned uint8 (B, 6, H, W) batches -> DetectionPipeline(u8=True).submit_u8 — the H2D copy
    runs on its own stream under the forwards in flight, `/255`, the RGB / IR split and the cast happen in the first kernel.  Reported
    beside `value`, never as it (the contract's `value` has its inputs resident in HBM)."""
    from icafusion_amd.pipeline import DetectionPipeline
    pipe = DetectionPipeline(model, B, H, W, dev, conf_thres=args.conf, iou_thres=args.iou, world=1, overlap=not args.no_overlap,
                             depth=args.depth, u8=True)
    nbuf = pipe.nplans + 2
    g = torch.Generator().manual_seed(1234)
    host = [torch.randint(0, 256, (B, 6, H, W), dtype=torch.uint8, generator=g).pin_memory() for _ in range(nbuf)]
    for k in range(max(args.warmup, nbuf)):
        pipe.submit_u8(host[k % nbuf])
    pipe.synchronize()
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(args.steps):
            pipe.submit_u8(host[k % nbuf])
        pipe.synchronize()
        torch.cuda.synchronize()
        rates.append(B * args.steps / (time.perf_counter() - t0))
    # the copy alone (no forward behind it): what the link gives this buffer size
    cs = torch.cuda.Stream(device=dev)
    dst = pipe.plans[0].inputs[0]

    def copy(k):
        with torch.cuda.stream(cs):
            dst.copy_(host[k % nbuf], non_blocking=True)
    copy(0)
    cs.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        copy(k)
    cs.synchronize()
    copy_s = (time.perf_counter() - t0) / 20
    nbytes = B * 6 * H * W
    rate = sorted(rates)[1]
    return {"pairs_per_s_with_h2d": round(rate, 2), "min": round(min(rates), 2), "max": round(max(rates), 2),
            "host_bytes_per_batch": nbytes, "pcie_gbs_achieved_in_loop": round(rate / B * nbytes / 1e9, 2),
            "pcie_gbs_copy_alone": round(nbytes / copy_s / 1e9, 2), "copy_alone_ms_per_batch": round(1e3 * copy_s, 3),
            "feed": "DMA engine (Tensor.copy_)",
            "streams": {"forward": [st.stream_id for st in pipe.fwd_streams], "nms": pipe.nms_stream.stream_id, "copy": [st.stream_id for st in pipe.copy_streams]},
            "note": f"pinned host uint8 (B,6,H,W) -> ONE copy on a high-priority copy stream straight into the input buffer of one of {pipe.nplans} plans "
                    f"(the one not in flight), {pipe.depth} batch(es) in flight, {nbuf} rotating host buffers; "
                    "forward from uint8 (icaf_stem2 / icaf_preprocess_u8) + NMS; median of 3 x K steps"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.dry_run:
        sys.exit(dry_run(args))
    from icafusion_amd import dist as D
    from icafusion_amd import ops
    from icafusion_amd.options import OPT
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_images, synth_state_dict
    from icafusion_amd.pipeline import DetectionPipeline
    import torch.distributed as tdist

    rank, world, local = D.init_from_env()
    if world != args.gpus:             # (a launcher's WORLD_SIZE wins; n_gpus in the JSON line is the world size that ran)
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    affinity = None
    if world > 1 and os.environ.get("ICAF_NO_PIN") != "1":      # one process per GPU: keep each rank's host threads next to its GPU
        pr = torch.cuda.get_device_properties(local)
        bus = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0" if hasattr(pr, "pci_bus_id") else None
        affinity = D.pin_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), bus)
        if affinity.get("pinned"):
            torch.set_num_threads(max(1, min(8, affinity["cpus"])))

    tag = {"kaist": "kaist", "FLIR": "FLIR", "VEDAI": "VEDAI", "LLVIP": "LLVIP"}[args.dataset]
    yaml_name = f"yolov5{args.model}_Transfusion_{tag}.yaml"
    cfg = yaml.safe_load(open(os.path.join(ROOT, "models", "transformer", yaml_name)))
    model = Model(cfg).eval()
    sd = synth_state_dict(model, seed=0)
    model.load_state_dict(sd)
    for i in (20, 21, 22):
        model.model[i].crosstransformer[0].loops = args.loops
        model.model[i].fuse_tail = not args.no_fuse_tail
    model = model.to(dev)
    model.compute_dtype = DT[args.dtype]
    model.static_outputs = True
    model.fold_upsample = args.fold_upsample
    model.autotune = not args.no_autotune
    B, H, W = args.batch, args.height, args.width
    strong = args.global_batch > 0
    if strong:                          # contiguous shards of a fixed global batch; uneven splits pad the short ranks to the longest shard
        lo, hi = D.shard_range(args.global_batch, rank, world)
        n_local, B = hi - lo, D.padded_local_batch(args.global_batch, world)
    else:
        n_local = B
    default_cache = os.path.join(ROOT, "profiles", "tune_cache.json")     # committed igemm tile choices: the same kernels
    if args.tune_cache and os.path.exists(args.tune_cache):               # run (and were profiled) from round to round;
        ops.load_tune_cache(args.tune_cache)                              # layers missing from it are tuned on the spot
    elif not args.tune_cache and os.path.exists(default_cache):
        ops.load_tune_cache(default_cache)
    model.use_graph = not args.no_graph
    if args.force_gather and world == 1 and not tdist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        tdist.init_process_group(backend="nccl", rank=0, world_size=1)
    pipe = DetectionPipeline(model, B, H, W, dev, conf_thres=args.conf, iou_thres=args.iou, world=world,
                             overlap=not args.no_overlap, depth=args.depth, force_gather=args.force_gather and world == 1)
    plan = pipe.plan
    if args.tune_cache and rank == 0:
        ops.save_tune_cache(args.tune_cache)
    # inputs resident in HBM before the timed region: each rank synthesises its own shard of the global batch
    rgb, ir = synth_images(B, H, W, seed=100 + rank)              # (strong scaling: rows >= n_local of a short rank are padding pairs)
    for pl in pipe.plans:
        pl.inputs[0].copy_(rgb.to(dev))
        pl.inputs[1].copy_(ir.to(dev))
    nc = cfg["nc"]
    sp = pipe.fwd_stream.cuda_stream
    torch.cuda.synchronize()
    step = pipe.step

    def barrier():
        if world > 1:
            tdist.barrier()

    for _ in range(args.warmup):
        step()
    # The timed region of the contract — barrier + synchronize, exactly K steps, synchronize + barrier, MAX over ranks — is
    # run `repeats` times back to back; `value` comes from the MEDIAN run.  One region is only K x ~2.6 ms long, and a single
    # sample of it moves by several per cent with the box's clocks / whatever else the host is doing.
    runs = []
    while len(runs) < max(1, args.repeats) or (sum(runs) < args.min_timed_seconds and len(runs) < 2000):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            det, count = step()[:2]
        pipe.synchronize()                      # (N > 1: also sends a gather group that K steps left incomplete)
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            el = float(t.item())
        runs.append(el)
    elapsed = sorted(runs)[len(runs) // 2]
    # per-rank rate (each rank's own clock around K of its steps, no barrier inside): the spread over ranks shows a slow GPU / link
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.synchronize()
    torch.cuda.synchronize()
    own = n_local * args.steps / (time.perf_counter() - t0)
    rank_rates = [own]
    if world > 1:
        t = torch.zeros((world,), dtype=torch.float64, device=dev)
        t[rank] = own
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
        rank_rates = [float(v) for v in t.tolist()]

    # ---- forward-only rate and per-kernel HIP-event timing (instrumented pass, outside the timed region) -------
    ev0, ev1 = ops.Event(), ops.Event()
    ev0.record(sp)
    for _ in range(args.steps):
        plan.run(sp)
    ev1.record(sp)
    fwd_ms = ev0.elapsed_ms(ev1) / args.steps              # ONE plan replayed back to back: the latency of a forward
    # forward-only THROUGHPUT with the run's number of batches in flight (no NMS, no gather)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pipe.plans[k % pipe.depth].run(pipe.fwd_streams[k % pipe.depth].cuda_stream)
    torch.cuda.synchronize()
    fwd_tp_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    # NMS alone on an idle GPU (in the timed region it overlaps the next forward, and its kernels then wait for free CUs)
    nsp = pipe.nms_stream.cuda_stream
    torch.cuda.synchronize()
    from icafusion_amd.utils.general import nms_device
    zs = plan.outputs[0]                                   # the predictions of the last forward of plan 0
    nms_device(zs, stream_ptr=nsp, runner=pipe.runners[0], **pipe.nms_args)
    n0, n1 = ops.Event(), ops.Event()
    n0.record(nsp)
    for _ in range(20):
        nms_device(zs, stream_ptr=nsp, runner=pipe.runners[0], **pipe.nms_args)
    n1.record(nsp)
    nms_ms = n0.elapsed_ms(n1) / 20
    saved_graph, plan.graph = plan.graph, None
    per_kernel = {}
    reps = 3
    for _ in range(reps):
        for l, (name, ms, flops, nbytes) in zip(plan.launches, plan.timed_run(sp)):
            kname = ops.conv_kernel_name(l) if l.fn is ops.lib().icaf_conv2d else name
            d = per_kernel.setdefault(kname, [0.0, 0.0, 0.0, 0])
            d[0] += ms; d[1] += flops; d[2] += nbytes; d[3] += 1
    total_ms = sum(v[0] for v in per_kernel.values()) / reps
    dom = max(per_kernel.items(), key=lambda kv: kv[1][0])
    dname, (dms, dflops, dbytes, dn) = dom
    # the same kernel's launch duration WITH a second forward in flight (what a rocprofv3 kernel trace of the default, overlapped run
    # shows): plan 1 replays as graphs on its own stream while plan 0 is timed launch by launch
    over_us = None
    if pipe.depth > 1:
        osp = pipe.fwd_streams[1].cuda_stream
        acc_ms, acc_n = 0.0, 0
        for _ in range(reps):
            for _ in range(12):
                pipe.plans[1].run(osp)
            for l, (name, ms, flops, nbytes) in zip(plan.launches, plan.timed_run(sp)):
                kname = ops.conv_kernel_name(l) if l.fn is ops.lib().icaf_conv2d else name
                if kname == dname:
                    acc_ms += ms; acc_n += 1
            torch.cuda.synchronize()
        over_us = round(1e3 * acc_ms / max(acc_n, 1), 2)
    plan.graph = saved_graph
    # which roof bounds the dominant kernel: its algorithmic intensity against the ridge of the chip (dense MFMA peak /
    # HBM peak = 312 FLOP/B for the 16-bit types).  Both fractions are reported.
    tflops = dflops / (dms * 1e-3) / 1e12 if dflops else 0.0
    gbs = dbytes / (dms * 1e-3) / 1e9
    intensity = dflops / dbytes if dbytes else 0.0
    ridge = PEAK_TFLOPS[args.dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    if dflops and intensity >= ridge:
        roof = {"bound": "mfma", "kernel": dname, "achieved": round(tflops, 2), "peak": PEAK_TFLOPS[args.dtype],
                "unit": "TFLOP/s", "frac": round(tflops / PEAK_TFLOPS[args.dtype], 4)}
    else:
        roof = {"bound": "hbm", "kernel": dname, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4)}
    roof.update({"intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                 "mfma_tflops": round(tflops, 2), "mfma_frac": round(tflops / PEAK_TFLOPS[args.dtype], 4),
                 "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)})
    roof.update({"traffic": None, "traffic_unit": "HBM bytes per launch (PMC, profiles/pmc_traffic*.json)",
                 "algorithmic_bytes_per_launch": round(dbytes / dn), "avg_launch_us": round(1e3 * dms / dn, 2),
                 "avg_launch_us_with_second_forward_in_flight": over_us, "launches_per_step": dn // reps,
                 "share_of_forward_kernel_time": round(dms / reps / total_ms, 3)})
    att = per_kernel.get("cross_attention")
    workload = (f"{yaml_name[:-5]} + DMFF(loops={args.loops}) {args.dtype}, batch {B}/GPU, "
                f"{H}x{W} synthetic RGB/IR pairs, seeded random weights, NMS conf {args.conf} iou {args.iou}")
    # SQ counters of the same command line (committed summary, see sq_counters): which pipe a kernel keeps busy.  `bound` = the roof the
    # dominant kernel sits closest to: HBM (algorithmic bytes / time / 8 TB/s), MFMA (pipe-busy fraction by the counters when there are
    # counters, FLOPs / time / dense peak otherwise) or VALU issue (fraction of the SIMDs' cycles in which a vector instruction issues:
    # SiLU's two transcendentals per value, address arithmetic, packing) — the front kernels of this model are VALU-bound, not HBM-bound
    sq, sq_src = sq_counters(workload)
    dsq = sq.get(dname, {})
    fr_hbm, fr_mfma = gbs / PEAK_HBM_GBS, dsq.get("mfma_util", tflops / PEAK_TFLOPS[args.dtype])
    by_roof = {"hbm": round(fr_hbm, 4), "mfma": round(fr_mfma, 4), "valu": dsq.get("valu_issue_frac")}
    roof["bound_by_intensity"] = roof["bound"]
    roof["bound"] = binding_roof(by_roof)
    roof.update({"frac_of_each_roof": by_roof, "valu_issue_frac": dsq.get("valu_issue_frac"), "mfma_busy": dsq.get("mfma_util"),
                 "wave_wait_frac": dsq.get("wave_wait_frac"), "valu_per_mfma": dsq.get("valu_per_mfma"), "sq_source": sq_src,
                 "note": "achieved / peak / frac are the byte (or FLOP) roofline of the contract: algorithmic work per launch over the HIP-event launch time; "
                         "`bound` names the largest of frac_of_each_roof (valu / mfma from the SQ counters of the same command line)"})
    # every kernel of the forward with its own two roofline fractions (the `roofline` object above is the first entry of this table: the kernel
    # with the largest share of the forward's kernel time)
    kernels = {k: {"ms_per_step": round(v[0] / reps, 4), "launches": v[3] // reps,
                   "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[1] else None,
                   "gbs": round(v[2] / (v[0] * 1e-3) / 1e9, 1),
                   "mfma_frac": round(v[1] / (v[0] * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4) if v[1] else None,
                   "hbm_frac": round(v[2] / (v[0] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                   "valu_issue_frac": sq.get(k, {}).get("valu_issue_frac"), "mfma_busy": sq.get(k, {}).get("mfma_util"),
                   "wave_wait_frac": sq.get(k, {}).get("wave_wait_frac")} for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])}
    for k, v in kernels.items():
        v["bound"] = binding_roof({"hbm": v["hbm_frac"], "mfma": v["mfma_busy"] if v["mfma_busy"] is not None else v["mfma_frac"], "valu": v["valu_issue_frac"]})
    # the whole forward against each pipe: every kernel's busy fraction weighted by its launch time (one batch at a time, the basis of the counters)
    kt = {k: v[0] / reps for k, v in per_kernel.items()}
    have = [k for k in kt if k in sq and sq[k].get("valu_issue_frac") is not None]
    pipe_busy = None
    if have:
        cov = sum(kt[k] for k in have)
        pipe_busy = {"valu_busy_us": round(1e3 * sum(kt[k] * sq[k]["valu_issue_frac"] for k in have), 1),
                     "mfma_busy_us": round(1e3 * sum(kt[k] * sq[k]["mfma_util"] for k in have), 1),
                     "wave_wait_frac": round(sum(kt[k] * (sq[k].get("wave_wait_frac") or 0.0) for k in have) / cov, 3),
                     "kernel_time_us": round(1e3 * total_ms, 1), "kernel_time_covered_by_counters": round(cov / total_ms, 3)}
    # DMFF block by SURVEY 8d's formula: (linear + bmm FLOPs of the block's kernels) / (their time x the dense MFMA peak)
    dm = [per_kernel[k] for k in ("dmff_ln_qkv", "cross_attention", "dmff_proj_mlp", "dmff_proj_mlp_reduce", "dmff_attn_mlp", "layernorm") if k in per_kernel]
    dmff_block = None
    if dm:
        dms_, dfl_ = sum(v[0] for v in dm) / reps, sum(v[1] for v in dm) / reps
        dmff_block = {"ms_per_step": round(dms_, 4), "gflop_per_step": round(dfl_ / 1e9, 2), "tflops": round(dfl_ / (dms_ * 1e-3) / 1e12, 1),
                      "mfma_frac": round(dfl_ / (dms_ * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                      "kernels": [k for k in ("dmff_ln_qkv", "cross_attention", "dmff_proj_mlp", "dmff_proj_mlp_reduce", "dmff_attn_mlp", "layernorm") if k in per_kernel],
                      "attention_mfma_busy": sq.get("cross_attention", {}).get("mfma_util"),
                      "note": "LN + QKV, crossed attention, out-proj + MLP (+ reduce) of all three levels and iterations; pooling / merge / the 1x1 fuse conv are not in it"}

    if rank == 0:
        pairs = (args.global_batch if strong else B * world) * args.steps
        value = pairs / elapsed
        gf = GFLOP_PER_PAIR.get((args.model, H, W))
        out = {
            "metric": "RGB/IR image-pairs/sec (two-stream forward + NMS)", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "repeats": len(runs), "timed_seconds": round(sum(runs), 3), "value_min": round(pairs / max(runs), 2), "value_max": round(pairs / min(runs), 2),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": args.global_batch if strong else B * world,
                       "local_batches": [D.shard_range(args.global_batch, r, world)[1] - D.shard_range(args.global_batch, r, world)[0]
                                         for r in range(world)] if strong else [B] * world,
                       "parallelism": f"dp{world} (pairs sharded, one all-gather of detections)",
                       "world_size_of_process_group": tdist.get_world_size() if world > 1 else 1,
                       "rank0_cpu_affinity": affinity,
                       "backend": tdist.get_backend() if world > 1 else None,
                       "graph": not args.no_graph, "nms_overlapped_with_next_forward": not args.no_overlap, "batches_in_flight": pipe.depth, "forced_one_rank_all_gather": bool(pipe.gather and world == 1),
                       "all_gather": {"steps_per_collective": pipe.group, "bytes_per_rank_per_collective": int(pipe.group_block.numel() * 4) if pipe.group > 1 else int(pipe.group_block.numel() * 4 // max(1, len(pipe.runners))),
                                      "stream": "own stream behind the group's last NMS"} if pipe.gather else None,
                       "fused_paths": pipe.plans[0].fusion_report(),
                       "plan_options": OPT.as_dict(), "plan_options_not_default": OPT.non_default()},
            "per_rank_pairs_per_s": {"min": round(min(rank_rates), 2), "max": round(max(rank_rates), 2),
                                     "note": "each rank's own clock around K steps of its shard (value = all ranks, max-over-ranks time)"},
            "forward_only_pairs_per_s": round(B / (fwd_tp_ms * 1e-3), 2),       # same batches-in-flight as `value`, no NMS
            "forward_ms_per_batch": round(fwd_ms, 3),                            # latency of ONE forward (one plan replayed back to back)
            "forward_only_pairs_per_s_one_in_flight": round(B / (fwd_ms * 1e-3), 2),
            "nms_ms_per_batch_standalone": round(nms_ms, 4),
            "model_tflops": round(gf * B / (fwd_tp_ms * 1e-3) / 1e3, 2) if gf else None,
            "forward_roofline": {      # whole forward: algorithmic FLOPs and leaf-op bytes of all launches over the forward-only time per batch
                "basis": f"forward-only throughput with {pipe.depth} batch(es) in flight",
                "tflops": round(sum(v[1] for v in per_kernel.values()) / reps / (fwd_tp_ms * 1e-3) / 1e12, 1),
                "mfma_frac": round(sum(v[1] for v in per_kernel.values()) / reps / (fwd_tp_ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                "gbs": round(sum(v[2] for v in per_kernel.values()) / reps / (fwd_tp_ms * 1e-3) / 1e9, 1),
                "hbm_frac": round(sum(v[2] for v in per_kernel.values()) / reps / (fwd_tp_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "pipe_busy_one_batch_at_a_time": pipe_busy},
            "roofline": roof,
            "dmff_block": dmff_block,
            "attention_kernel": None if att is None else {
                "tflops": round(att[1] / (att[0] * 1e-3) / 1e12, 2), "ms_per_step": round(att[0] / reps, 4),
                "mfma_frac_of_dense_peak": round(att[1] / (att[0] * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4)},
            "kernels": kernels,
            "plan_buffer_MB": round(plan.nbytes / 2 ** 20, 1),
            "detections_first_image": int(count.reshape(-1)[0]),
        }
        if pipe_busy:
            fr = out["forward_roofline"]
            fr["valu_issue_frac"] = round(pipe_busy["valu_busy_us"] / pipe_busy["kernel_time_us"], 4)
            fr["mfma_busy"] = round(pipe_busy["mfma_busy_us"] / pipe_busy["kernel_time_us"], 4)
            fr["bound"] = binding_roof({"hbm": fr["hbm_frac"], "mfma": fr["mfma_busy"], "valu": fr["valu_issue_frac"]})
        if world == 1 and not args.no_latency:
            # batch-1 latency (detect_twostream.py:83-88 runs forward -> sync -> NMS -> sync per frame pair): one pair, one plan, the
            # hipGraph replay followed by NMS on the same stream, host-synchronised every step; median of 100
            lp = DetectionPipeline(model, 1, H, W, dev, conf_thres=args.conf, iou_thres=args.iou, world=1, overlap=False, depth=1)
            r1, i1 = synth_images(1, H, W, seed=7)
            lp.inputs[0].copy_(r1.to(dev)); lp.inputs[1].copy_(i1.to(dev))
            lat, lat_f = [], []
            for k in range(120):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                lp.step()
                lp.synchronize()
                if k >= 20:
                    lat.append(1e3 * (time.perf_counter() - t0))
            l0, l1 = ops.Event(), ops.Event()
            lsp = lp.fwd_stream.cuda_stream
            l0.record(lsp)
            for _ in range(50):
                lp.plan.run(lsp)
            l1.record(lsp)
            torch.cuda.synchronize()
            lat.sort()
            out["latency_ms_b1"] = round(lat[len(lat) // 2], 4)                        # forward + NMS, host clock around one step
            out["latency_b1"] = {"forward_plus_nms_ms_median": round(lat[len(lat) // 2], 4), "p90_ms": round(lat[int(len(lat) * 0.9)], 4),
                                 "forward_graph_ms_device": round(l0.elapsed_ms(l1) / 50, 4), "launches_per_forward": len(lp.plan.launches),
                                 "note": "batch 1, depth 1, hipGraph replay + device NMS on one stream, synchronised per step"}
        if world == 1 and not args.no_h2d:
            out["h2d_feed"] = h2d_feed(model, args, B, H, W, dev)
        # PMC counters cannot be read from inside this process: the committed summaries of tools/gpu_pmc.sh (same command line, one
        # file per workload: profiles/pmc_traffic*.json) supply the dominant kernel's HBM bytes
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_traffic*.json"))):
            pm = json.load(open(tf))
            pk = pm.get("kernels", {})
            k = pk.get(dname)
            if pm.get("workload") == out["config"]["workload"] and k \
                    and "fetch_bytes_corrected" in k and "write_bytes_uncorrected" in k:
                roof["traffic"] = round(k["fetch_bytes_corrected"] + k["write_bytes_uncorrected"])
                roof["traffic_detail"] = {"fetch_bytes": round(k["fetch_bytes_corrected"]),
                                          "write_bytes": round(k["write_bytes_uncorrected"]), "source": os.path.basename(tf)}
                # whole forward: counter bytes per dispatch of every kernel name x its launches per forward (a name's counter mean is over
                # exactly these launches, so the product is the name's exact total); names without counters are listed, not guessed
                tot, missing = 0.0, []
                for name, v in per_kernel.items():
                    c = pk.get(name)
                    if c and "fetch_bytes_corrected" in c:
                        tot += (c["fetch_bytes_corrected"] + c["write_bytes_uncorrected"]) * (v[3] // reps)
                    else:
                        missing.append(name)
                fr = out["forward_roofline"]
                fr["traffic"] = round(tot)
                fr["traffic_unit"] = "HBM bytes per forward (PMC bytes per dispatch x launches, all kernels)"
                fr["traffic_gbs"] = round(tot / (fwd_tp_ms * 1e-3) / 1e9, 1)
                fr["traffic_hbm_frac"] = round(tot / (fwd_tp_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                fr["traffic_over_algorithmic"] = round(tot / (sum(v[2] for v in per_kernel.values()) / reps), 3)
                fr["traffic_kernels_without_counters"] = missing
                break
        if world == 1 and not args.no_cpu_baseline:
            fused = Model(cfg).eval()
            fused.load_state_dict(sd)
            out["cpu_baseline"] = cpu_baseline(cfg, fused.fuse().state_dict(), args, args.loops)
        # ONE JSON line, and the LAST thing on stdout: RCCL prints a version banner through C stdio when its first communicator comes up
        # (N > 1, --force-gather); flush that buffer first, or it lands after (or inside) the line when the process exits
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if tdist.is_initialized():
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
