"""CPU, world_size 2 over gloo: the batch-sharding plumbing of dffinthewild_amd.dist (the HIP forward
itself cannot run without a GPU, so a deterministic stand-in with the Network call signature takes
its place; what is under test is the partition, the rank order of the gather and ragged shards)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dffinthewild_amd import dist as ddist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_model(FS, fd):
    # per-stack "depth" = mean of the stack broadcast to (H,W): identifies which stack produced which map
    B, _, N, H, W = FS.shape
    m = FS.mean(dim=(1, 2, 3, 4)).reshape(B, 1, 1).expand(B, H, W).contiguous()
    return m, m, m, m + fd.reshape(B, -1)[:, :1].reshape(B, 1, 1) * 0


def _fake_e2e_model(FS, fd, fovs):
    # End_to_End signature: 5 outputs, the FOVs take part so that a mis-sliced third input would show
    m, _, _, p3 = _fake_model(FS, fd)
    return m, m, m, p3 + fovs.reshape(FS.shape[0], -1)[:, :1].reshape(-1, 1, 1), FS


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = ddist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    s, e = ddist.shard_bounds(total, world, rank)
    H = W = 4
    full = torch.arange(total, dtype=torch.float32).reshape(total, 1, 1, 1, 1).expand(total, 3, 2, H, W).contiguous()
    fd = torch.ones(total, 2, 1, 1)
    outs, gathered = ddist.sharded_depth(_fake_model, full[s:e], fd[s:e], total=total)
    assert outs[3].shape[0] == e - s
    # End_to_End variant: a third per-sample input, sliced like the others
    fov = 100.0 * torch.arange(total, dtype=torch.float32).reshape(total, 1, 1, 1, 1).expand(total, 1, 2, 1, 1).contiguous()
    outs5, g5 = ddist.sharded_depth(_fake_e2e_model, full[s:e], fd[s:e], fov[s:e], total=total)
    assert len(outs5) == 5 and g5[:, 0, 0].tolist() == [101.0 * i for i in range(total)]
    # shard sizes not announced (total=None): the sizes are settled by one small collective first, equal and ragged alike
    again = ddist.all_gather_depth(outs[3])
    assert torch.equal(again, gathered)
    # ... and sizes that are not the contiguous split raise on every rank instead of hanging in a mismatched collective
    bad = outs[3][:1] if rank == 0 else torch.cat([outs[3], outs[3]])[: (e - s) + 1]
    with pytest.raises(ValueError, match="contiguous split"):
        ddist.all_gather_depth(bad)
    q.put((rank, gathered[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_two_rank_shard_and_gather(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, vals in got:
        assert vals == [float(i) for i in range(total)], (rank, vals)


def test_shard_bounds_cover_everything():
    for total in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [ddist.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_passthrough():
    x = torch.rand(3, 4, 4)
    assert ddist.all_gather_depth(x) is x
    assert ddist.env_world()[2] >= 1
