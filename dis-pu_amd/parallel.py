"""Patch-level data parallelism: the reference has no distributed code at all; patches are
independent, so the batch of patches is split contiguously over the ranks (one process per GPU, no
data-path collective) and ONE all-gather reassembles the upsampled clouds on every rank
(`torch.distributed` backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous [lo, hi) of `n_items` for `rank`: the first (n_items % world) ranks get one extra item."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_rows(local, world, group):
    """[r, ...] (same r on every rank) -> [world * r, ...].  One all_gather_into_tensor (RCCL: a single collective of
    r * row_bytes per rank); device tensors under gloo (ranks sharing a GPU in tests) hop through host memory."""
    local = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host = local.detach().cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return torch.cat(parts, dim=0).to(local.device)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def shard_sizes(local_rows, n_items, group=None):
    """Rows held by every rank.  With `n_items` they follow from shard_bounds (no communication); without it the
    ranks exchange their row counts (one tiny all-gather), so ragged shards never masquerade as equal ones."""
    world = dist.get_world_size(group)
    if n_items is not None:
        sizes = [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]
        if sizes[dist.get_rank(group)] != int(local_rows):
            raise ValueError("this rank holds %d rows but shard_bounds(%d, rank %d, world %d) says %d"
                             % (local_rows, n_items, dist.get_rank(group), world, sizes[dist.get_rank(group)]))
        return sizes
    mine = torch.tensor([int(local_rows)], dtype=torch.int64)
    if dist.get_backend(group) != "gloo":
        mine = mine.cuda()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return [int(p.item()) for p in parts]


def all_gather_clouds(local, n_items=None, out=None, group=None):
    """local [b_local, M, 3] -> [n_items, M, 3] on every rank, in global patch order.
    Equal shards use one all_gather_into_tensor (a single ring/direct collective of b_local*M*12 bytes per
    rank); ragged shards are padded to the largest shard and trimmed."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(local.shape[0], n_items, group)
    if len(set(sizes)) == 1:
        if out is not None and not (local.is_cuda and dist.get_backend(group) == "gloo"):
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
            return out
        res = _all_gather_rows(local, world, group)
        if out is not None:
            out.copy_(res)
            return out
        return res
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = _all_gather_rows(pad, world, group)
    res = torch.cat([buf[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)
    if out is not None:
        out.copy_(res)
        return out
    return res


def upsample_sharded(forward, patches, group=None):
    """Run `forward(local_patches) -> local_clouds` on this rank's contiguous shard of `patches`
    ([n_items, N, 3], identical on every rank) and return all clouds [n_items, M, 3] on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return forward(patches)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(patches.shape[0], rank, world)
    return all_gather_clouds(forward(patches[lo:hi]), n_items=patches.shape[0], group=group)


def all_reduce_gradients(flat, group=None):
    """Replica data parallelism of the training step (BASELINE config 5): ONE all-reduce(sum) of the
    flat gradient bucket (every trainable variable, 4.2 MB fp32) per step.  Returns the world size; the 1/world
    average is applied by the caller (folded into the Adam launch).  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1:
        return 1
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        host = flat.detach().cpu()                       # smoke-test configuration only (ranks sharing a GPU)
        dist.all_reduce(host, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, group=group)
    return world


def average_replica_stats(tensors, group=None):
    """BatchNorm moving statistics are updated from per-rank batch statistics (the reference is single-GPU); averaging
    the 2 x 16 floats after each step keeps the replicas bit-identical."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    for t in tensors:
        if t.is_cuda and dist.get_backend(group) == "gloo":
            host = t.detach().cpu()
            dist.all_reduce(host, group=group)
            t.copy_(host / world)
        else:
            dist.all_reduce(t, group=group)
            t.div_(world)


# ------------------------------------------------------------------------------------------------------------------
# Collectives off the compute stream (round 5).  Both data-path collectives are latency-bound (393 KB of clouds per rank, a
# 4.2 MB gradient bucket) against a 0.9 / 1.7 ms step; issued on the compute stream after the step they are fully exposed.
#   * GatherPipeline: the all-gather of step i runs on a side stream while step i + 1 computes (double-buffered results);
#   * BucketedAllReduce: the gradient buffer is reduced in contiguous buckets, each launched as soon as the backward pass has
#     produced it (the refine branch's gradients are final half a backward before the feature extractor's).
# Transport: RCCL (backend "nccl") -> a side HIP stream ordered against compute by events, the launching thread returns at
# once.  gloo (CPU tests, ranks sharing a GPU) has no stream semantics: ONE background thread drains a FIFO of jobs (FIFO =
# the same collective order on every rank) and stages device tensors through host memory, so the launching thread also
# returns at once and the GPU keeps computing.
# ------------------------------------------------------------------------------------------------------------------
import queue
import threading


class _Ticket(object):
    __slots__ = ("flag", "event", "error")

    def __init__(self):
        self.flag, self.event, self.error = threading.Event(), None, None


class _Lane(object):
    """Asynchronous collectives, issued strictly in submission order.

    Failure model: a job that raises on one rank never enters its collective, so the peers would sit in the matching call for
    the transport's timeout.  The lane therefore breaks loudly: the failing rank tears its lane group down (the peers' pending
    operations then fail with a connection error instead of hanging), every later submit / wait on this lane re-raises the first
    error, the lane's own gloo group carries a finite timeout (`timeout_s`) as the backstop, and close() raises if the background
    thread cannot be drained instead of abandoning it."""

    def __init__(self, device, group=None, timeout_s=300.0, threaded=None):
        """threaded: None = the transport's default (gloo: a background thread, it has no streams; RCCL: enqueue from the launching
        thread).  True under RCCL moves the ~30 us of host time every collective call costs (c10d + RCCL enqueue) off the thread that
        launches the step's kernels -- worth it for launch-bound steps (the 8-patch train step); the lane then owns a communicator."""
        self.group = group
        self.backend = dist.get_backend(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.threaded = (self.backend == "gloo") if threaded is None else bool(threaded)
        self.timeout_s = float(timeout_s)
        self.broken = None                      # the first exception of a job: the lane is unusable afterwards
        self._own_group = False
        self._q, self._thread = None, None
        if self.threaded:
            # the background thread's collectives must not interleave with the launching thread's on one gloo context (the pairing
            # of sends and receives is by call order): the lane talks over a group of its own.  (RCCL needs none: there every
            # collective, the lane's included, is ENQUEUED by the launching thread, in program order on every rank.)
            # dist.new_group is itself a collective over the parent group: lanes are built where every rank passes in the same
            # order (GatherPipeline / BucketedAllReduce constructors; Trainer builds its reducer when the parameters are loaded).
            import datetime
            # (a threaded RCCL lane needs its own communicator for the same reason: two threads enqueueing on one communicator race.)
            self.group = dist.new_group(ranks=dist.get_process_group_ranks(group if group is not None else dist.group.WORLD), backend=self.backend,
                                        timeout=datetime.timedelta(seconds=self.timeout_s))
            self._own_group = True
            self._q = queue.Queue()
            self._thread = threading.Thread(target=self._run, name="dispu-comm", daemon=True)
            self._thread.start()

    # ---- the collectives themselves (device tensors hop through host memory under gloo) ----
    def all_gather(self, out, local):
        world = dist.get_world_size(self.group)
        if local.is_cuda and self.backend == "gloo":
            host = local.detach().cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host, group=self.group)
            out.copy_(torch.cat(parts, dim=0))
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)

    def all_reduce(self, t, scale=None):
        if t.is_cuda and self.backend == "gloo":
            host = t.detach().cpu()
            dist.all_reduce(host, group=self.group)
            t.copy_(host if scale is None else host * scale)
        else:
            dist.all_reduce(t, group=self.group)
            if scale is not None:
                t.mul_(scale)

    # ---- failure ----
    def _abort(self, err):
        """first failure on this rank: remember it and drop the lane's own group, so that peers blocked in the collective this rank
        never entered fail at once (connection closed) instead of waiting for the timeout."""
        if self.broken is None:
            self.broken = err
            if self._own_group:
                self._own_group = False
                try:
                    dist.destroy_process_group(self.group)
                except Exception:              # noqa: BLE001 -- already failing; the original error is the one to report
                    pass

    def _check(self):
        if self.broken is not None:
            raise RuntimeError("the collective lane of this rank is broken by an earlier failure: %s: %s"
                               % (type(self.broken).__name__, self.broken)) from self.broken

    # ---- submission ----
    def submit(self, job, after=()):
        """`job()` issues collectives; it starts once the HIP events in `after` have completed.  Returns a ticket for wait()."""
        self._check()
        ticket = _Ticket()
        if self.threaded:
            self._q.put((job, tuple(after), ticket))
            return ticket
        with torch.cuda.stream(self.stream):
            for ev in after:
                self.stream.wait_event(ev)
            try:
                job()                              # RCCL: enqueued behind the waits; the lane stream is blocked until it completes
            except BaseException as e:             # noqa: BLE001
                self._abort(e)
                raise
            ticket.event = torch.cuda.Event()
            ticket.event.record(self.stream)
        ticket.flag.set()
        return ticket

    def _run(self):
        if self.cuda:
            torch.cuda.set_device(self.device)
        while True:
            item = self._q.get()
            if item is None:
                return
            job, after, ticket = item
            if self.broken is not None:            # jobs behind a failed one are not attempted: their collectives have no partner
                ticket.error = self.broken
                ticket.flag.set()
                continue
            try:
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        for ev in after:
                            if self.backend == "gloo":
                                ev.synchronize()   # host-staged transport: blocks this thread only
                            else:
                                self.stream.wait_event(ev)     # RCCL: enqueued behind the producers, nothing blocks
                        job()
                        ticket.event = torch.cuda.Event()
                        ticket.event.record(self.stream)
                else:
                    job()
            except BaseException as e:             # noqa: BLE001 -- re-raised in the thread that waits for the ticket
                ticket.error = e
                self._abort(e)
            ticket.flag.set()

    def wait(self, ticket):
        """the CURRENT stream (and, under gloo, the calling thread) waits for the job behind `ticket`."""
        if ticket is None:
            return
        if not ticket.flag.wait(timeout=self.timeout_s + 30.0 if self.threaded else None):
            err = TimeoutError("a collective job did not finish within %.0f s" % self.timeout_s)
            self._abort(err)
            raise err
        if ticket.error is not None:
            raise ticket.error
        if ticket.event is not None:
            torch.cuda.current_stream(self.device).wait_event(ticket.event)

    def close(self):
        if self._thread is not None:
            self._q.put(None)
            self._thread.join(timeout=self.timeout_s + 30.0)
            alive, self._thread = self._thread.is_alive(), None
            if alive:
                err = RuntimeError("the collective lane's thread is still inside a collective after %.0f s" % self.timeout_s)
                self._abort(err)
                raise err


def _here(device):
    """an event at the current position of the current stream (None for CPU tensors: the data is there when the call returns)."""
    if torch.device(device).type != "cuda":
        return ()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return (ev,)


class GatherPipeline(object):
    """Reassembly of the upsampled clouds (BASELINE config 3) overlapped with the next step's compute.

        pipe = GatherPipeline((b_local, M, 3), device)
        slot, buf = pipe.acquire()      # the step writes its clouds into `buf` (Generator.fine_out = buf)
        ... launch the step ...
        pipe.launch(slot)               # all-gather of `buf` on the comm lane, ordered after everything queued so far
        ... next step (other slot) ...
        clouds = pipe.result(slot)      # [world * b_local, M, 3]; the current stream waits for that gather only

    `depth` result buffers: a slot is reused `depth` steps later, and acquire() makes the current stream wait for the gather that
    last used it (finished long before -- it never stalls a steady-state loop).  Equal shards only (the sharded bench / serving
    loop); ragged batches go through all_gather_clouds."""

    def __init__(self, local_shape, device, dtype=torch.float32, depth=2, group=None, threaded=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GatherPipeline needs an initialised process group")
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.lane = _Lane(self.device, group, threaded=threaded)
        local_shape = tuple(int(s) for s in local_shape)
        self.local = [torch.empty(local_shape, dtype=dtype, device=self.device) for _ in range(depth)]
        self.out = [torch.empty((self.world * local_shape[0],) + local_shape[1:], dtype=dtype, device=self.device) for _ in range(depth)]
        self.tickets = [None] * depth
        self.steps = 0

    def acquire(self):
        s = self.steps % len(self.local)
        self.steps += 1
        self.lane.wait(self.tickets[s])
        return s, self.local[s]

    def launch(self, s):
        loc, out, lane = self.local[s], self.out[s], self.lane
        self.tickets[s] = lane.submit(lambda: lane.all_gather(out, loc), after=_here(self.device))

    def result(self, s):
        self.lane.wait(self.tickets[s])
        return self.out[s]

    def drain(self):
        for t in self.tickets:
            self.lane.wait(t)

    def close(self):
        self.drain()
        self.lane.close()


class BucketedAllReduce(object):
    """sum-all-reduce of one flat buffer in contiguous buckets, each launched when ITS producers are done.

    `bounds` = [(lo, hi), ...] element ranges that tile `flat`, in the order the backward pass completes them.  launch(i, after)
    queues bucket i on the comm lane behind the HIP events `after` (default: the current position of the current stream);
    finish() launches whatever was not launched, makes the current stream wait for every bucket and returns the world size (the
    1/world average is the caller's: Trainer folds it into the Adam launch).  Summation order inside a bucket is the
    transport's, exactly as for the single-bucket all_reduce_gradients: same values, sooner."""

    def __init__(self, flat, bounds, group=None, threaded=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("BucketedAllReduce needs an initialised process group")
        bounds = [(int(lo), int(hi)) for lo, hi in bounds]
        cover = sorted(bounds)
        if cover[0][0] != 0 or cover[-1][1] != flat.numel() or any(cover[i][1] != cover[i + 1][0] for i in range(len(cover) - 1)):
            raise ValueError("buckets %r do not tile the %d-element buffer" % (bounds, flat.numel()))
        self.flat, self.bounds, self.group = flat, bounds, group
        self.world = dist.get_world_size(group)
        self.lane = _Lane(flat.device, group, threaded=threaded)
        self.tickets = [None] * len(bounds)

    def launch(self, i, after=None):
        if self.tickets[i] is not None:
            raise RuntimeError("bucket %d launched twice in one step" % i)
        lo, hi = self.bounds[i]
        view, lane = self.flat[lo:hi], self.lane
        self.tickets[i] = lane.submit(lambda: lane.all_reduce(view), after=_here(self.flat.device) if after is None else after)

    def launched(self, i):
        return self.tickets[i] is not None

    def finish(self, extra=()):
        """`extra`: small tensors AVERAGED over the replicas behind the last bucket (BatchNorm moving statistics)."""
        for i in range(len(self.bounds)):
            if self.tickets[i] is None:
                self.launch(i)
        tickets, self.tickets = self.tickets, [None] * len(self.bounds)
        if extra:
            lane, inv = self.lane, 1.0 / self.world
            tickets.append(lane.submit(lambda: [lane.all_reduce(t, scale=inv) for t in extra], after=_here(self.flat.device)))
        for t in tickets:
            self.lane.wait(t)
        return self.world

    def close(self):
        self.lane.close()
