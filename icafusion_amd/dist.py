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


def shard_range(global_batch, rank, world):
    """Contiguous split of the global batch; the first (global_batch % world) ranks take one extra pair."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_detections(det, count):
    """(B, max_det, 6) fp32 + (B,) int32 -> one (B, max_det*6 + 1) fp32 block (count stored exactly as a float)."""
    B = det.shape[0]
    return torch.cat((det.reshape(B, -1), count.to(torch.float32).reshape(B, 1)), 1).contiguous()


def unpack_detections(block, max_det):
    B = block.shape[0]
    det = block[:, :max_det * 6].reshape(B, max_det, 6)
    count = block[:, max_det * 6].round().to(torch.int32)
    return det, count


def gather_detections(det, count, out=None, force_collective=False):
    """All-gather every rank's detections (equal B_local on all ranks).  Returns (det_all, count_all) on every rank,
    ordered by rank, i.e. in global batch order for a contiguous shard.  With one rank nothing needs exchanging and the
    inputs are returned — unless force_collective is set, which sends the block through the collective anyway (a
    1-GPU box can then exercise the RCCL call, its stream ordering and the gathered buffer: tests/test_gpu_model.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        if force_collective:
            raise RuntimeError("gather_detections(force_collective=True) needs an initialised process group")
        return det, count
    if dist.get_world_size() == 1 and not force_collective:
        return det, count
    world = dist.get_world_size()
    block = pack_detections(det, count)
    if out is None:
        out = torch.empty((world * block.shape[0], block.shape[1]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block)
    return unpack_detections(out, det.shape[1])
