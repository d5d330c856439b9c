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
