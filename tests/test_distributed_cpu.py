"""World-size-2 `gloo` test (CPU) of the N > 1 path: contiguous patch sharding + all-gather reassembly
(dis-pu_amd/parallel.py, the code bench.py uses with backend "nccl" on GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(p):            # stands in for the generator: deterministic, per-patch independent, 4x points
    return torch.cat([p, p * 2 + 1, p - 3, p * p], dim=1)


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dispu_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        patches = torch.rand(n_items, 8, 3, generator=g)
        out = parallel.upsample_sharded(_fake_forward, patches)
        lo, hi = parallel.shard_bounds(n_items, rank, world)
        q.put((rank, lo, hi, torch.equal(out, _fake_forward(patches)), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])
def test_sharded_upsample_gloo_world2(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_items     # contiguous cover
    assert all(r[3] for r in res) and all(r[4] == (n_items, 32, 3) for r in res)


def test_shard_bounds_cover():
    from dispu_amd import parallel
    for n in (1, 7, 32, 256, 257):
        for w in (1, 2, 4, 8):
            b = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


# ---- training step: replica data parallelism (BASELINE config 5) -------------------------------------------
def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dispu_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # a least-squares "model": loss = mean over the batch of (x.w - y)^2; global batch 8 split contiguously
        g = torch.Generator().manual_seed(3)
        X, y, w = torch.randn(8, 5, generator=g, dtype=torch.float64), torch.randn(8, generator=g, dtype=torch.float64), torch.randn(5, generator=g, dtype=torch.float64)
        lo, hi = parallel.shard_bounds(8, rank, world)
        r = X[lo:hi] @ w - y[lo:hi]
        flat = (2.0 * X[lo:hi].t() @ r / (hi - lo)).clone()              # gradient of this rank's mean loss
        n = parallel.all_reduce_gradients(flat)
        full = 2.0 * X.t() @ (X @ w - y) / 8                             # gradient of the global-batch mean loss
        stats = [torch.full((16,), float(rank)), torch.full((16,), 10.0 + rank)]
        parallel.average_replica_stats(stats)
        q.put((rank, n, bool(torch.allclose(flat / n, full, rtol=1e-12, atol=1e-12)), float(stats[0][0]), float(stats[1][0])))
    finally:
        dist.destroy_process_group()


def test_gradient_all_reduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2, 2]
    assert all(r[2] for r in res)                          # averaged shard gradients == global-batch gradient
    assert all(r[3] == 0.5 and r[4] == 10.5 for r in res)  # BN moving statistics averaged over the replicas


def test_all_reduce_is_noop_without_process_group():
    from dispu_amd import parallel
    t = torch.arange(4.0)
    assert parallel.all_reduce_gradients(t) == 1 and torch.equal(t, torch.arange(4.0))


# ---- round 5: collectives off the critical path (parallel.GatherPipeline / BucketedAllReduce) ----------------------------
def _pipeline_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dispu_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, m = 3, 16
        pipe = parallel.GatherPipeline((b, m, 3), "cpu")
        ok, pending = True, []

        def produce(step, r):                                 # what rank r's "generator" writes at `step`
            g = torch.Generator().manual_seed(1000 * step + r)
            return torch.rand(b, m, 3, generator=g)

        def check(step, slot):
            want = torch.cat([produce(step, r) for r in range(world)], dim=0)
            return torch.equal(pipe.result(slot), want) and torch.equal(pipe.result(slot), parallel.all_gather_clouds(produce(step, rank)))

        for step in range(7):                                 # consume step i only AFTER step i + 1 was launched (the overlap pattern)
            slot, buf = pipe.acquire()
            buf.copy_(produce(step, rank))
            pipe.launch(slot)
            if pending:
                ok = ok and check(*pending.pop())
            pending.append((step, slot))
        ok = ok and check(*pending.pop())
        pipe.close()

        # gradient buckets: [60, 100) first (the part the backward pass finishes first), then [0, 60); stats averaged behind them
        g = torch.Generator().manual_seed(50 + rank)
        flat = torch.randn(100, generator=g, dtype=torch.float64)
        want = sum(torch.randn(100, generator=torch.Generator().manual_seed(50 + r), dtype=torch.float64) for r in range(world))
        ref = flat.clone()
        parallel.all_reduce_gradients(ref)
        ar = parallel.BucketedAllReduce(flat, [(60, 100), (0, 60)])
        stats = [torch.full((16,), float(rank)), torch.full((16,), 10.0 + rank)]
        ar.launch(0)
        early = ar.launched(0) and not ar.launched(1)
        n = ar.finish(extra=stats)
        ok2 = n == world and torch.equal(flat, want) and torch.equal(flat, ref) and early
        flat.copy_(torch.randn(100, generator=torch.Generator().manual_seed(50 + rank), dtype=torch.float64))
        n = ar.finish()                                       # second step, nothing launched early: everything goes out in finish()
        ok2 = ok2 and torch.equal(flat, want)
        try:
            parallel.BucketedAllReduce(flat, [(0, 50), (60, 100)])
            ok2 = False
        except ValueError:
            pass
        ar.close()
        q.put((rank, ok, ok2, float(stats[0][0]), float(stats[1][0])))
    finally:
        dist.destroy_process_group()


def test_overlapped_collectives_gloo_world2():
    """GatherPipeline: seven steps through two result slots, each result read one step late -> equal to the blocking
    all_gather_clouds of the same step.  BucketedAllReduce: two buckets in backward-completion order == the single-bucket
    all-reduce, replica statistics averaged behind them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "pipelined all-gather differs from the blocking one"
    assert all(r[2] for r in res), "bucketed all-reduce differs from the single-bucket one"
    assert all(r[3] == 0.5 and r[4] == 10.5 for r in res)


# ---- round 6: a failing job on one rank must not hang its peers ------------------------------------------------------------
def _failing_lane_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dispu_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lane = parallel._Lane("cpu", timeout_s=20.0)
        t = torch.ones(4)
        good = lane.submit(lambda: lane.all_reduce(t))
        lane.wait(good)
        ok_first = bool(torch.equal(t, torch.full((4,), float(world))))

        def job():
            if rank == 1:
                raise ValueError("rank 1's producer failed before the collective")
            lane.all_reduce(t)

        t0 = time.time()
        kind = None
        try:
            lane.wait(lane.submit(job))
        except ValueError as e:
            kind = "own:" + str(e)
        except Exception as e:                    # noqa: BLE001 -- the peer: a transport error, not a hang
            kind = "peer:" + type(e).__name__
        waited = time.time() - t0
        later = None
        try:
            lane.submit(lambda: None)
        except RuntimeError as e:
            later = str(e)
        try:
            lane.close()
        except RuntimeError:
            pass
        q.put((rank, ok_first, kind, waited, later))
    finally:
        dist.destroy_process_group()


def test_lane_failure_is_loud_on_every_rank_gloo_world2():
    """a job that raises on one rank never enters its collective.  The failing rank re-raises its own error and
    tears its lane group down; the peer's pending all-reduce then fails with a transport error well inside the lane's timeout (20 s
    here) instead of hanging; afterwards the lane refuses further work on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_lane_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[1][2] == "own:rank 1's producer failed before the collective"
    assert res[0][2] is not None and res[0][2].startswith("peer:"), res[0]
    assert res[0][3] < 60.0 and res[1][3] < 60.0, "a rank sat in the dead collective: %r" % (res,)
    assert all(r[4] is not None and "broken" in r[4] for r in res), res
