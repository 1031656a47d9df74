"""Batch sharding of focal stacks over the GPUs of one node (one process per GPU).

The reference's only multi-GPU mechanism is ``nn.DataParallel`` (test.py:32): scatter the batch on
dim 0, replicate the weights, gather the outputs.  The forward has no cross-sample operation
(eval-mode BatchNorm), so the MI355X-native form is: every rank owns a contiguous slice of the
batch and its own copy of the packed weights, runs the HIP engine on its slice, and the per-rank
depth maps are collected with ONE all-gather (RCCL over xGMI when the backend is ``nccl``; gloo on
CPU in the tests).  There is no collective on the data path before that.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init_process_group(backend=None, set_device=True):
    """Join the job's process group (no-op for a single process).  ``nccl`` is RCCL on ROCm."""
    rank, local_rank, world = env_world()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl" and set_device:
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(total, world, rank):
    """Contiguous [start, stop) of ``total`` stacks owned by ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_depth(local, total=None):
    """Collect per-rank depth maps (b_r, H, W) into (sum b_r, H, W) on every rank, in rank order.

    Equal shards use a single ``all_gather_into_tensor`` (one RCCL ncclAllGather on the compute
    stream); ragged shards are padded to the largest shard first.  ``total`` = global number of
    stacks (needed only for ragged shards)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    b = local.shape[0]
    if local.is_cuda and dist.get_backend() == "gloo":
        # test configuration only (several ranks sharing one GPU, no RCCL): gloo gathers host tensors
        return all_gather_depth(local.cpu(), total).to(local.device)
    if total is None:
        # shard sizes unknown: one small collective settles them (a mismatched all_gather_into_tensor would hang or
        # return garbage instead of failing)
        sizes = torch.zeros(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(sizes, torch.tensor([b], dtype=torch.int64, device=local.device))
        sizes = sizes.tolist()
        total = int(sum(sizes))
        expect = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
        if sizes != expect:
            raise ValueError(f"all_gather_depth: per-rank batch sizes {sizes} are not the contiguous split {expect} of "
                             f"{total} stacks that shard_bounds() deals out")
    if total == b * world:
        out = local.new_empty((world * b,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    bmax = -(-total // world)
    padded = local.new_zeros((bmax,) + tuple(local.shape[1:]))
    padded[:b] = local
    out = local.new_empty((world * bmax,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded)
    parts = []
    for r in range(world):
        s, e = shard_bounds(total, world, r)
        parts.append(out[r * bmax: r * bmax + (e - s)])
    return torch.cat(parts, 0)


def sharded_depth(model, FS_local, fd_local, *extra_local, total=None, gather=True):
    """Run ``model`` (a dffinthewild_amd Network, or any callable with its signature) on this rank's
    slice of the batch and return (local output tuple, gathered pred3 or None).  ``extra_local`` are further
    per-sample inputs sliced the same way — the FOVs of the End_to_End variant: ``model(FS, fd, FOVs)``."""
    outs = model(FS_local, fd_local, *extra_local)
    gathered = all_gather_depth(outs[3], total) if gather else None
    return outs, gathered
