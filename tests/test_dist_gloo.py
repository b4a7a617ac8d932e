"""world_size-2 gloo tests (CPU) of the N>1 host path: gradient averaging, Sync-BN statistics
exchange, batch sharding."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pixelssl_amd import dist as pdist
    pdist.init_from_env(backend="gloo")
    assert pdist.is_distributed() and pdist.world_size() == world and pdist.rank() == rank
    torch.manual_seed(0)
    full = torch.randn(8, 16, 5, 5)                    # the "global" batch every rank can reconstruct
    mine = full[rank * 4:(rank + 1) * 4]
    # (1) gradient exchange: mean of per-rank mean-gradients == global-batch gradient
    w = torch.ones(16, requires_grad=True)
    (mine * w.view(1, -1, 1, 1)).pow(2).mean().backward()
    g = w.grad.clone()
    pdist.allreduce_mean_(g)
    wg = torch.ones(16, requires_grad=True)
    (full * wg.view(1, -1, 1, 1)).pow(2).mean().backward()
    ok_grad = torch.allclose(g, wg.grad, rtol=1e-5, atol=1e-7)
    # (2) Sync-BN statistics: all-reduced [sum, sumsq] reproduce the global-batch mean/var
    stats = torch.cat([mine.sum(dim=(0, 2, 3)), (mine ** 2).sum(dim=(0, 2, 3))])
    pdist.allreduce_sum_(stats)
    n = full.numel() / 16
    mean = stats[:16] / n
    var = stats[16:] / n - mean ** 2
    ok_bn = torch.allclose(mean, full.mean(dim=(0, 2, 3)), atol=1e-5) and \
        torch.allclose(var, full.var(dim=(0, 2, 3), unbiased=False), atol=1e-5)
    q.put((rank, bool(ok_grad), bool(ok_bn)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_and_syncbn_exchange():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res


def test_batch_sharding_rules():
    from pixelssl_amd import dist as pdist
    assert pdist.shard_batch_sizes(32, 32, 8) == (4, 4)
    with pytest.raises(ValueError):
        pdist.shard_batch_sizes(6, 4, 4)
    assert pdist.world_size() == 1 and pdist.rank() == 0 and not pdist.is_distributed()
