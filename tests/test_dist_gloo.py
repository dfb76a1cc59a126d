"""world_size-2 gloo test of the sharding + detection all-gather (CPU, no GPU needed)."""
import os
import socket

import pytest

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
    assert det.shape == (world, gb // world, max_det, 6) and cnt.shape == (world, gb // world) and cnt.dtype == torch.int32
    ok = all(torch.equal(x, y) for x, y in zip(D.flatten_gathered(det, cnt), (det_all, cnt_all)))
    # the serving path: NMS output lives in ONE pre-allocated block, which is what the collective sends (no cat, no cast)
    block, d, c = D.detection_block(b - a, max_det, "cpu")
    d.copy_(det_all[a:b]); c.copy_(cnt_all[a:b])
    out = torch.empty((world * block.numel(),))
    det2, cnt2 = D.gather_detections(d, c, out=out, block=block)
    ok = ok and det2.data_ptr() == out.data_ptr() and all(torch.equal(x, y) for x, y in zip(D.flatten_gathered(det2, cnt2), (det_all, cnt_all)))
    # the pipeline's grouped form: the blocks of `group` consecutive steps lie side by side and travel in ONE collective (DetectionPipeline._issue_gather)
    group, Bl = 3, b - a
    blk = Bl * max_det * 6 + Bl
    gblock = torch.zeros((group * blk,))
    for sidx in range(group):
        _, d_s, c_s = None, gblock[sidx * blk:sidx * blk + Bl * max_det * 6].view(Bl, max_det, 6), gblock[sidx * blk + Bl * max_det * 6:(sidx + 1) * blk].view(torch.int32)
        d_s.copy_(det_all[a:b] + 10.0 * sidx); c_s.copy_((cnt_all[a:b] + sidx) % (max_det + 1))
    gout = torch.empty((world * group * blk,))
    torch.distributed.all_gather_into_tensor(gout, gblock)
    for sidx in range(group):
        dg, cg = D.split_group_block(gout, world, group, sidx, Bl, max_det)
        df, cf = D.flatten_gathered(dg, cg)
        ok = ok and dg.shape == (world, Bl, max_det, 6) and cg.dtype == torch.int32 and torch.equal(df, det_all + 10.0 * sidx) \
            and torch.equal(cf, (cnt_all + sidx) % (max_det + 1))
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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start two ranks itself (it silently ran one and reported
    n_gpus 1).  --dry-run stops after the process group + one detection-block all-gather (gloo here, RCCL on a GPU node)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["requested_gpus"] == 2 and rec["gather_ok"] and rec["backend"] == "gloo"


def test_strong_scaling_helpers_cover_uneven_splits():
    """--global-batch: the longest shard sizes the static plan; gathered_to_global drops the padding rows of the shorter ranks and
    restores global pair order (BASELINE configs 3 / 5 split evenly: 256 -> 8 x 32, 128 -> 8 x 16; 10 over 4 ranks does not)."""
    from icafusion_amd import dist as D
    assert D.padded_local_batch(256, 8) == 32 and D.padded_local_batch(128, 8) == 16 and D.padded_local_batch(10, 4) == 3
    G, world, max_det = 10, 4, 2
    bpad = D.padded_local_batch(G, world)
    det = torch.full((world, bpad, max_det, 6), -1.0)
    cnt = torch.full((world, bpad), -1, dtype=torch.int32)
    for r in range(world):
        lo, hi = D.shard_range(G, r, world)
        for k in range(hi - lo):
            det[r, k] = float(lo + k)
            cnt[r, k] = lo + k
    dg, cg = D.gathered_to_global(det, cnt, G)
    assert dg.shape == (G, max_det, 6) and cg.tolist() == list(range(G)) and [float(dg[i, 0, 0]) for i in range(G)] == list(range(G))
    det_even, cnt_even = det[:, :2].contiguous(), cnt[:, :2].contiguous()       # (world, 2, ...): 8 pairs over 4 ranks, no padding
    de, ce = D.gathered_to_global(det_even, cnt_even, 8)                       # an even split comes back as VIEWS of the gathered block
    assert de.shape == (8, max_det, 6) and ce.shape == (8,)
    assert de.untyped_storage().data_ptr() == det_even.untyped_storage().data_ptr() and de.data_ptr() == det_even.data_ptr()
    assert ce.untyped_storage().data_ptr() == cnt_even.untyped_storage().data_ptr() and ce.data_ptr() == cnt_even.data_ptr()
    assert torch.equal(de, det_even.reshape(8, max_det, 6)) and torch.equal(ce, cnt_even.reshape(8))


def test_bench_strong_scaling_launch_with_four_ranks_and_an_uneven_split():
    """`python bench.py --gpus 4 --global-batch 10 --dry-run`: four gloo ranks, shards of 3 / 3 / 2 / 2 pairs padded to 3, ONE
    all-gather of equally sized blocks, the global view holds exactly the 10 pairs in order."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "4", "--global-batch", "10", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 4 and rec["gather_ok"] and rec["backend"] == "gloo" and rec["global_batch"] == 10
    assert rec["local_batches"] == [3, 3, 2, 2]


CONFIG3 = ["--model", "l", "--global-batch", "256"]                                                            # BASELINE configs[2]: yolov5l bf16, batch 256 over 8 GPUs
CONFIG5 = ["--model", "l", "--dataset", "VEDAI", "--dtype", "f16", "--height", "1280", "--width", "1280", "--global-batch", "128", "--conf", "0.3"]   # configs[4]


@pytest.mark.parametrize("name,cfg_args,shard", [("config3", CONFIG3, 32), ("config5", CONFIG5, 16)])
def test_bench_world8_dry_run_of_the_eight_gpu_baseline_configurations(name, cfg_args, shard):
    """The exact command lines of BASELINE configs 3 and 5 at eight ranks (gloo here, RCCL on a node), `--dry-run`: every rank starts, binds, joins the
    process group, takes its contiguous shard (256 -> 8 x 32, 128 -> 8 x 16: even, no padding), ONE all-gather of the detection blocks, the global view
    holds the pairs in order.  Both launch forms: bench.py spawning its own ranks, and the driver's `python -m torch.distributed.run ... bench.py`."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    bench = os.path.join(repo, "bench.py")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    forms = {"self-spawn": [sys.executable, bench, "--gpus", "8"] + cfg_args + ["--dry-run"],
             "torchrun": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), bench, "--gpus", "8"] + cfg_args + ["--dry-run"]}
    for form, cmd in forms.items():
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (name, form, r.stdout[-1000:] + r.stderr[-3000:])
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert rec["n_gpus"] == 8 and rec["requested_gpus"] == 8 and rec["gather_ok"] and rec["backend"] == "gloo", (name, form, rec)
        assert rec["global_batch"] == shard * 8 and rec["local_batches"] == [shard] * 8, (name, form, rec)
