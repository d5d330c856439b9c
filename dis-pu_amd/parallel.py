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
