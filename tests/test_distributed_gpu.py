"""The N > 1 code paths with the REAL generator / Trainer (BASELINE configs[2] and [4] in miniature): two ranks sharing
the one GPU of the test box over `gloo` (bench.py's DISPU_BENCH_BACKEND=gloo configuration; the driver's multi-GPU runs use
nccl == RCCL, one GPU per rank, through exactly the same dis-pu_amd/parallel.py calls).

  * inference: patches sharded contiguously, each rank runs the generator on its shard, ONE all-gather reassembles
    the clouds -> bit-identical to the unsharded forward on every rank (DisPU/model.py:333-339 loop, batched);
  * training: replica data parallelism, 2 x 4 patches: the all-reduced gradient bucket equals the sum of the two
    shard gradients computed without a process group, parameters after Adam are bit-identical on both ranks, and the
    step agrees with ONE 8-patch step (DisPU/model.py:215-232) up to the per-rank BatchNorm batch statistics.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _infer_worker(rank, world, port, n_items, q):
    _init(rank, world, port)
    try:
        from dispu_amd import parallel, synth
        from dispu_amd.generator import Generator
        from dispu_amd.params import init_params
        dev = torch.device("cuda:0")
        gen = Generator(params=init_params(seed=1234), device=dev)
        x = torch.from_numpy(synth.patches(n_items, 256, seed=31)).to(dev)
        calls = []

        def forward(p):
            calls.append(int(p.shape[0]))
            return gen(p)[1].clone()

        out = parallel.upsample_sharded(forward, x)
        full = gen(x)[1]
        torch.cuda.synchronize()
        lo, hi = parallel.shard_bounds(n_items, rank, world)
        q.put((rank, calls, (lo, hi), bool(torch.equal(out, full)), tuple(out.shape), float(out.abs().sum())))
    finally:
        dist.destroy_process_group()


def _spawn(target, args, world=2, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=timeout))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("n_items", [8, 7])
def test_real_generator_sharded_equals_unsharded(dev, n_items):
    res = _spawn(_infer_worker, (n_items,))
    assert [r[0] for r in res] == [0, 1]
    for rank, calls, (lo, hi), same, shape, _ in res:
        assert calls == [hi - lo], "rank %d ran the generator on %r patches, shard is %d" % (rank, calls, hi - lo)
        assert same, "rank %d: sharded + all-gather differs from the unsharded forward" % rank
        assert shape == (n_items, 1024, 3)
    assert res[0][5] == res[1][5]                       # both ranks hold the same gathered clouds


def _train_worker(rank, world, port, q):
    _init(rank, world, port)
    try:
        from dispu_amd import parallel, synth
        from dispu_amd.params import init_params
        from dispu_amd.train import Trainer
        dev = torch.device("cuda:0")
        P = init_params(seed=1234)
        x, gt = synth.patch_with_gt(8, 256, 1024, seed=41)
        x, gt = torch.from_numpy(x).to(dev), torch.from_numpy(gt).to(dev)
        radius = torch.ones(8, device=dev)

        def local_grads(lo, hi):                        # forward + loss + backward of one shard, no collective
            t = Trainer(params=P, device=dev)
            t.zero_grad()
            t.forward(x[lo:hi])
            t.loss_backward(gt[lo:hi], radius[lo:hi])
            t.backward()
            torch.cuda.synchronize()
            return t.flat_g.clone(), t

        g0, _ = local_grads(0, 4)
        g1, _ = local_grads(4, 8)
        g8, _ = local_grads(0, 8)

        lo, hi = parallel.shard_bounds(8, rank, world)
        tr = Trainer(params=P, device=dev)              # default process group = the 2-rank gloo group
        tr.zero_grad()
        tr.forward(x[lo:hi])
        terms = tr.loss_backward(gt[lo:hi], radius[lo:hi])
        tr.backward()
        n = tr.all_reduce_grads()
        summed = tr.flat_g.clone()
        tr.adam(n)
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).norm() / b.norm())
        # a second, full train_step (the public entry point) to make sure it runs under the group as well
        t2 = tr.train_step(x[lo:hi], gt[lo:hi], radius[lo:hi])
        torch.cuda.synchronize()
        q.put((rank, n, rel(summed, g0 + g1), rel(summed / n, g8), tr.flat_p.cpu().numpy().tobytes(),
               tr.moving_mean.cpu().numpy().tobytes(), float(terms["pu_loss"]), float(t2["pu_loss"])))
    finally:
        dist.destroy_process_group()


def test_real_trainer_data_parallel_step(dev):
    res = _spawn(_train_worker, ())
    for rank, n, rel_sum, rel_full, _, _, loss, loss2 in res:
        assert n == 2
        # float atomics in the scatter gradients make two evaluations of one shard differ in the last bits
        assert rel_sum <= 1e-5, "rank %d: all-reduced bucket vs sum of shard gradients: %g" % (rank, rel_sum)
        # vs ONE 8-patch step: identical up to the BatchNorm batch statistics (per-rank here)
        assert rel_full <= 2e-2, "rank %d: DP 2x4 vs one 8-patch step: %g" % (rank, rel_full)
        assert np.isfinite(loss) and np.isfinite(loss2)
    assert res[0][4] == res[1][4], "parameters differ between the replicas after the step"
    assert res[0][5] == res[1][5], "BN moving statistics differ between the replicas"
    print("DP check: bucket-vs-shards %.2e / %.2e, vs 8-patch step %.2e / %.2e" % (res[0][2], res[1][2], res[0][3], res[1][3]))
