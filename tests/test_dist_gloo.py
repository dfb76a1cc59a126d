"""world_size-2 gloo test of the sharding + detection all-gather (CPU, no GPU needed)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from icafusion_amd import dist as D
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    gb, max_det = 6, 5
    a, b = D.shard_range(gb, rank, world)
    g = torch.Generator().manual_seed(0)
    det_all = torch.rand((gb, max_det, 6), generator=g)
    cnt_all = torch.tensor([0, 5, 2, 3, 1, 4], dtype=torch.int32)
    det, cnt = D.gather_detections(det_all[a:b].clone(), cnt_all[a:b].clone())
    ok = torch.equal(det, det_all) and torch.equal(cnt, cnt_all)
    q.put((rank, ok, (a, b)))
    torch.distributed.destroy_process_group()


def test_shard_and_gather_world2():
    from icafusion_amd.dist import shard_range
    assert [shard_range(7, r, 3) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]
    assert [shard_range(256, r, 8)[1] - shard_range(256, r, 8)[0] for r in range(8)] == [32] * 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True, (0, 3)), (1, True, (3, 6))]
