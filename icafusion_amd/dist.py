"""Multi-GPU inference sharding: one process per GPU, image pairs split contiguously, ONE collective.

Eval-mode inference has no cross-sample operation (BN folded, LayerNorm / softmax per token, NMS per image —
SURVEY.md §8e), so the only exchange is an all-gather of the fixed-size detection blocks at the end:
per rank [B_local, max_det, 6] fp32 + [B_local] counts (~230 KB at B_local = 32), latency-bound on xGMI.
`torch.distributed` with backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun-style env vars; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(local, world_local, pci_bus_id=None, sysfs="/sys"):
    """Bind this rank's host threads to the CPUs next to its GPU (one process per GPU on a 256-thread, multi-socket host: a rank whose launch
    thread wanders to the far socket pays a cross-socket hop per hipGraphLaunch / RCCL proxy wake-up).  The GPU's NUMA node comes from sysfs
    (`/sys/bus/pci/devices/<bus id>/numa_node` -> `/sys/devices/system/node/node<N>/cpulist`); when the platform does not say (-1, no sysfs,
    a container without the files) the CPUs this process may use are dealt to the local ranks in equal contiguous slices instead.  Ranks whose
    GPUs sit on the same node share that node's CPUs.  Returns a dict describing what was done (bench.py logs it)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:                                  # not Linux
        return {"pinned": False, "reason": "no sched_getaffinity"}
    info = {"pinned": False, "numa_node": None}
    cpus = None
    if pci_bus_id:
        try:
            bid = pci_bus_id.lower()
            if bid.count(":") == 1:
                bid = "0000:" + bid
            node = int(open(os.path.join(sysfs, "bus/pci/devices", bid, "numa_node")).read())
            if node >= 0:
                node_cpus = _parse_cpulist(open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")).read()) & set(allowed)
                if node_cpus:
                    cpus, info["numa_node"] = sorted(node_cpus), node
        except (OSError, ValueError):
            pass
    if cpus is None:                                        # equal contiguous slices of whatever this process may run on
        n = len(allowed)
        cpus = allowed[local * n // world_local:(local + 1) * n // world_local] or allowed
        info["reason"] = "GPU NUMA node unknown: equal slices of the allowed CPUs"
    try:
        os.sched_setaffinity(0, cpus)
        info.update(pinned=True, cpus=len(cpus), first_cpu=cpus[0], last_cpu=cpus[-1])
    except OSError as e:
        info["reason"] = f"sched_setaffinity: {e}"
    return info


def shard_range(global_batch, rank, world):
    """Contiguous split of the global batch; the first (global_batch % world) ranks take one extra pair."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def padded_local_batch(global_batch, world):
    """Pairs per rank of a STATIC plan that can hold any rank's shard: the longest shard, ceil(global_batch / world).  The collective
    needs equally sized blocks, so a rank whose shard is one pair shorter runs that slot as padding (gathered_to_global drops it)."""
    return -(-global_batch // world)


def gathered_to_global(det_all, count_all, global_batch):
    """Rank-major gathered blocks (world, B_pad, max_det, 6) / (world, B_pad) of a contiguously sharded global batch -> the
    (global_batch, max_det, 6) / (global_batch,) tensors in global pair order: rank r contributes its first shard_range(r) rows.
    Even splits (global_batch % world == 0) return views; uneven ones allocate (index_select)."""
    world, bpad = count_all.shape[0], count_all.shape[1]
    if global_batch == world * bpad:
        return flatten_gathered(det_all, count_all)
    rows = []
    for r in range(world):
        lo, hi = shard_range(global_batch, r, world)
        rows.extend(r * bpad + k for k in range(hi - lo))
    idx = torch.tensor(rows, dtype=torch.long, device=count_all.device)
    det, count = flatten_gathered(det_all, count_all)
    return det.index_select(0, idx), count.index_select(0, idx)


def detection_block(B, max_det, device):
    """ONE allocation holding a rank's NMS output: [B][max_det][6] fp32 detections followed by [B] int32 counts (stored in the
    same fp32 storage, bit for bit).  The NMS kernels write straight into the two views, and the block is what travels through the
    collective: no packing kernel, no allocation per step.  Returns (block, det view, count view)."""
    n = B * max_det * 6
    block = torch.zeros((n + B,), dtype=torch.float32, device=device)
    return block, block[:n].view(B, max_det, 6), block[n:].view(torch.int32)


def split_block(flat, world, B, max_det):
    """Views into `world` concatenated detection blocks: det (world, B, max_det, 6) fp32, count (world, B) int32 - rank-major,
    i.e. global batch order for contiguous shards (flatten_gathered gives the (world * B, ...) tensors, allocating)."""
    n = B * max_det * 6
    blocks = flat.view(world, n + B)
    return blocks[:, :n].view(world, B, max_det, 6), blocks[:, n:].view(torch.int32)


def split_group_block(flat, world, group, slot, B, max_det):
    """Views into a gathered GROUP: every rank sent `group` consecutive detection blocks in one collective (DetectionPipeline gathers once per
    `group` steps), so `flat` is (world, group, block); returns step `slot`'s det (world, B, max_det, 6) fp32 and count (world, B) int32."""
    n = B * max_det * 6
    blocks = flat.view(world, group, n + B)[:, slot]
    return blocks[:, :n].view(world, B, max_det, 6), blocks[:, n:].view(torch.int32)


def flatten_gathered(det, count):
    return det.reshape(-1, det.shape[-2], 6), count.reshape(-1)


def gather_detections(det, count, out=None, force_collective=False, block=None):
    """All-gather every rank's detections (equal B_local on all ranks): ONE `all_gather_into_tensor` of the rank's detection block.
    Returns (det_all (world, B_local, max_det, 6), count_all (world, B_local)) on every rank - views of `out`, rank-major.
    `block` = the rank's detection_block (det / count are its views: nothing is packed, nothing allocated when `out` is given);
    without it the two tensors are packed first (tests, ad-hoc callers).  With one rank nothing needs exchanging and the inputs
    come back as (1, B, ...) views - unless force_collective is set, which sends the block through the collective anyway (a 1-GPU
    box can then exercise the RCCL call, its stream ordering and the gathered buffer: tests/test_gpu_pipeline.py)."""
    B, max_det = det.shape[0], det.shape[1]
    active = dist.is_available() and dist.is_initialized()
    if not active and force_collective:
        raise RuntimeError("gather_detections(force_collective=True) needs an initialised process group")
    if not active or (dist.get_world_size() == 1 and not force_collective):
        return det.view(1, B, max_det, 6), count.view(1, B)
    world = dist.get_world_size()
    if block is None:
        block = torch.cat((det.reshape(-1), count.to(torch.int32).view(torch.float32).reshape(-1)))
    if out is None:
        out = torch.empty((world * block.numel(),), dtype=torch.float32, device=block.device)
    dist.all_gather_into_tensor(out.view(-1), block)
    return split_block(out.view(-1), world, B, max_det)
