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


def _foreign_worker(rank, world, port, q):
    """FusedSGD over plain torch parameters on two ranks where rank 1 has NO gradient for one of them (a rank-dependent branch
    of a plug-in model): both ranks must step that parameter with the mean (the rank without a gradient used to
    throw the averaged slice away and the replicas drifted apart), and a parameter without a gradient anywhere stays untouched."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      PXL_FORCE_DEVICE="cpu")
    from pixelssl_amd import dist as pdist
    from pixelssl_amd.nn.optimizer import FusedSGD
    pdist.init_from_env(backend="gloo")
    pdist.epoch_barrier()                                  # (the explicit epoch-boundary collective: both ranks pass it)
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(5))
    b = torch.nn.Parameter(torch.randn(3))
    c = torch.nn.Parameter(torch.randn(2))
    a0, b0, c0 = a.detach().clone(), b.detach().clone(), c.detach().clone()
    opt = FusedSGD([dict(params=[a, b, c])], lr=0.5, momentum=0.0, weight_decay=0.0)
    a.grad = torch.full((5,), float(rank + 1))             # ranks 1.0 / 2.0 -> mean 1.5
    if rank == 0:
        b.grad = torch.full((3,), 4.0)                     # rank 1 has none -> mean of (4, 0) = 2
    opt.step()
    ok = torch.allclose(a.detach(), a0 - 0.5 * 1.5) and torch.allclose(b.detach(), b0 - 0.5 * 2.0) and torch.equal(c.detach(), c0)
    # both ranks hold the same parameters afterwards
    flat = torch.cat([a.detach(), b.detach(), c.detach()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(g, gathered[0]) for g in gathered)
    q.put((rank, bool(ok), bool(same), c.grad is None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_optimizer_steps_a_partially_missing_gradient_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_foreign_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[3] for r in res), res
