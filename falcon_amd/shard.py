"""Multi-GPU sharding of a verify batch.

Every (public key, message, signature) triple is independent, so a batch of `count` items is cut
into `world` contiguous ranges of ceil(count / world) items, one per rank / GPU, and nothing is
exchanged on the data path. The only step that crosses ranks is the host-side gather of the
per-item accept bytes, done here with torch.distributed on CPU tensors over gloo (an RCCL-only
process group gets a gloo companion for it): no collective touches the GPUs or xGMI.

The same split is used by falcon::verify_many_sharded<N> in include/falcon_amd.hpp.
"""
import numpy as np

_CPU_GROUPS = {}   # process group -> its gloo companion (gather_accept)


def shard_range(count, world, rank):
    """Half-open item range [lo, hi) of `rank` among `world` ranks; empty for trailing ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    per = -(-count // world)
    lo = min(per * rank, count)
    return lo, min(lo + per, count)


def split_mixed(logn):
    """Index arrays of the Falcon-512 and Falcon-1024 items of a mixed batch (logn[i] in {9, 10});
    each class is then sharded on its own so every GPU gets the same N mix."""
    logn = np.asarray(logn)
    if not np.isin(logn, (9, 10)).all():
        raise ValueError("logn entries must be 9 or 10")
    return np.flatnonzero(logn == 9), np.flatnonzero(logn == 10)


def gather_accept(local_accept, count, dst=0, group=None):
    """Concatenate every rank's accept bytes on rank `dst` (returns None elsewhere).

    local_accept: uint8 array holding this rank's shard_range(count, world, rank) slice.
    Works without torch.distributed initialised (single process: returns the input)."""
    import torch
    import torch.distributed as dist

    local_accept = np.ascontiguousarray(local_accept, np.uint8)
    if not (dist.is_available() and dist.is_initialized()):
        if local_accept.shape[0] != count:
            raise ValueError("single process: local shard must be the whole batch")
        return local_accept
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(count, world, rank)
    if local_accept.shape[0] != hi - lo:
        raise ValueError(f"rank {rank}: shard holds {local_accept.shape[0]} items, expected {hi - lo}")
    per = -(-count // world)
    # A HOST-side gather (BASELINE.json north_star: "only a host-side gather of per-signature accept bitmaps remains"): CPU
    # tensors over gloo. A process group that is RCCL-only gets a gloo companion over the same ranks, made once (every rank
    # comes through here, so the collective new_group call is matched); nothing crosses xGMI.
    cpu_group = group
    if dist.get_backend(group) == "nccl":
        key = id(group) if group is not None else None
        if key not in _CPU_GROUPS:
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            _CPU_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo")
        cpu_group = _CPU_GROUPS[key]
    buf = torch.zeros(max(per, 1), dtype=torch.uint8)   # equal-sized pieces for all_gather
    buf[:hi - lo] = torch.from_numpy(local_accept)
    pieces = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(pieces, buf, group=cpu_group)
    if rank != dst:
        return None
    out = np.empty(count, np.uint8)
    for r in range(world):
        a, b = shard_range(count, world, r)
        out[a:b] = pieces[r][:b - a].numpy()
    return out


def verify_many_sharded(verify_fn, logn, pk, msg, sig, group=None, dst=0):
    """Run `verify_fn(logn, pk, msg, sig) -> accept` on this rank's contiguous shard of the batch and
    gather the accept bytes on rank `dst`. Every rank passes the same full (count, ...) arrays (or
    views of them; only the shard's rows are touched). verify_fn is Context.verify_many in
    production."""
    import torch.distributed as dist

    count = pk.shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(count, world, rank)
    local = verify_fn(logn, pk[lo:hi], msg[lo:hi], sig[lo:hi]) if hi > lo else np.zeros(0, np.uint8)
    return gather_accept(local, count, dst=dst, group=group)
