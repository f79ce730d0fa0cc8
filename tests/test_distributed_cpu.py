"""world_size-2 gloo tests of the data-parallel plumbing (CPU): flat-gradient all-reduce, shard bounds."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dposer_amd import distributed as ddp
    rk, ws, _ = ddp.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, world) and ddp.world_size() == world and ddp.rank() == rank
    g = torch.full((1000,), float(rank + 1))
    n = ddp.all_reduce_sum_(g)
    gb = torch.arange(100.0) * (rank + 1)
    seen = []
    nb = ddp.all_reduce_buckets_(gb, [(60, 100), (30, 60), (0, 30)], None)
    assert nb == world and torch.equal(gb, torch.arange(100.0) * 3)
    # per-batch metric dicts, unequal number of batches per rank (drop_last=False shards)
    mine = [{"mpjpe": torch.full((4,), float(rank)), "mpvpe": np.full(4, 10.0 * rank)} for _ in range(1 + rank)]
    means, allv = ddp.gather_metrics(mine)
    if rank == 0:
        assert len(allv["mpjpe"]) == 4 * 3 and abs(means["mpjpe"] - (0 * 4 + 1 * 8) / 12) < 1e-12 and abs(means["mpvpe"] - 80 / 12) < 1e-12
    else:
        assert means is None and allv is None
    # the same means from (sum, count) pairs with ONE all-reduce: every rank gets them, no per-sample values travel
    red = ddp.reduce_metric_means(mine)
    assert abs(red["mpjpe"] - 8 / 12) < 1e-12 and abs(red["mpvpe"] - 80 / 12) < 1e-12
    # local_only(): inside a data-parallel job the enclosed work runs as a single-process job would (bench.py's N = 1 leg of the same build)
    assert ddp.dp_active()
    with ddp.local_only():
        assert not ddp.dp_active()
        with ddp.local_only():
            assert not ddp.dp_active()
        assert not ddp.dp_active()
        h = torch.full((8,), float(rank + 1))
        assert ddp.all_reduce_sum_(h) == 1 and float(h[0]) == rank + 1            # no collective: untouched, "world" of one
    assert ddp.dp_active()
    flat = torch.arange(10.0) * (rank + 1)
    ddp.broadcast_(flat, src=0)
    lo, hi = ddp.shard_bounds(65536 + 3, world, rank)
    out.put((rank, n, float(g[0]), float(flat.sum()), lo, hi))
    ddp.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_and_shards():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2, 2]
    assert [r[2] for r in res] == [3.0, 3.0]                  # 1 + 2 on both ranks
    assert [r[3] for r in res] == [45.0, 45.0]                # rank 0's buffer everywhere
    assert (res[0][4], res[0][5], res[1][4], res[1][5]) == (0, 32770, 32770, 65539)


def test_single_process_is_noop():
    from dposer_amd import distributed as ddp
    g = torch.ones(8)
    assert ddp.all_reduce_sum_(g) == 1 and float(g.sum()) == 8.0
    assert ddp.all_reduce_buckets_(g, [(0, 4), (4, 8)]) == 1 and float(g.sum()) == 8.0
    means, allv = ddp.gather_metrics([{"e": [1.0, 3.0]}, {"e": torch.tensor([5.0])}])
    assert means == {"e": 3.0} and allv["e"] == [1.0, 3.0, 5.0]
    assert ddp.reduce_metric_means([{"e": [1.0, 3.0]}, {"e": torch.tensor([5.0])}]) == {"e": 3.0}
    assert ddp.shard_bounds(10, 3, 0) == (0, 4) and ddp.shard_bounds(10, 3, 2) == (7, 10)


def _worker_shards(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dposer_amd import distributed as ddp
    ddp.init_from_env(backend="gloo")
    n = 1003
    bounds = ddp.zero1_bounds(n, world)
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    ddp.reduce_scatter_flat_(g, bounds)
    lo, hi = bounds[rank]
    ok = torch.equal(g[lo:hi], torch.arange(n, dtype=torch.float32)[lo:hi] * 3)      # 1x + 2x
    p = torch.zeros(n)
    p[lo:hi] = float(rank + 1)
    ddp.all_gather_flat_(p, bounds)
    want = torch.cat([torch.full((b - a,), float(r + 1)) for r, (a, b) in enumerate(bounds)])
    # the same reduce-scatter range by range as the backward pass announces them (ZeRO-1 under the backward): ranges that straddle the
    # ownership boundary are cut there, an un-announced range is caught by finish(expect=...)
    g2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = ddp.StreamedReduceToOwners(g2, None, bounds)
    red.on_final([(700, 1003)], None)
    red.on_final([(400, 700), (100, 400)], None)
    caught = False
    try:
        red.finish(expect=[(0, 100)])
    except RuntimeError:
        caught = True
    red.on_final([(0, 100)], None)
    red.finish(expect=[(0, 100), (100, 400), (400, 700), (700, 1003)])
    ok = ok and caught and torch.equal(g2[lo:hi], torch.arange(n, dtype=torch.float32)[lo:hi] * 3) and len(red.works) == 5
    out.put((rank, bool(ok), bool(torch.equal(p, want)), bounds))
    ddp.barrier()
    dist.destroy_process_group()


def test_reduce_scatter_and_all_gather_of_flat_ranges():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shards, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)
    assert res[0][3] == [(0, 504), (504, 1003)]                     # 16-byte aligned shard starts, the last rank takes the rest


def _worker_autograd_dp(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from dposer_amd import distributed as ddp
    from dposer_amd.algorithms.advanced import losses, sde_lib
    ddp.init_from_env(backend="gloo")
    torch.manual_seed(0)                                     # identically seeded ranks
    w = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4))]
    w[0].grad = torch.full((5,), float(rank + 1))
    w[1].grad = torch.arange(6.0).reshape(2, 3) * (rank + 1)
    losses._dp_average_grads(w)                              # w[2] has no gradient on either rank
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    t, z = losses._dp_draws(sde, torch.zeros(16, 63))
    out.put((rank, w[0].grad.tolist(), w[1].grad.tolist(), w[2].grad is None, t.tolist(), float(z.abs().sum())))
    ddp.barrier()
    dist.destroy_process_group()


def test_two_rank_autograd_step_helpers():
    """The data-parallel pieces of the non-fused training steps (losses.aux_step / generic fallback): mean all-reduce of ``p.grad``
    and per-rank (t, z) draws although both ranks seeded torch identically."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_autograd_dp, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == [1.5] * 5
    assert res[0][2] == res[1][2] == (np.arange(6.0).reshape(2, 3) * 1.5).tolist()
    assert res[0][3] and res[1][3]
    assert res[0][4] != res[1][4] and all(1e-5 <= v <= 1.0 for v in res[0][4] + res[1][4])
    assert res[0][5] != res[1][5]
