"""world_size-2 tests of the data-parallel helpers on CPU with the gloo backend."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "so-net_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import torch.nn as nn
    from sonet_hip import dp
    w, r, _ = dp.init_distributed(backend="gloo")
    assert (w, r) == (world, rank) and dp.world_size() == world
    # shards: contiguous, disjoint, cover the batch
    lo, hi = dp.shard_range(7, world, rank)
    ranges = [None] * world
    dist.all_gather_object(ranges, (lo, hi))
    assert ranges == [(0, 4), (4, 7)]
    batch = {"pc": torch.arange(7 * 3).float().view(7, 3), "label": torch.arange(7)}
    mine = dp.shard_batch(batch, world, rank)
    assert mine["label"].tolist() == list(range(lo, hi))
    # parameter broadcast from rank 0 (the reference's init is unseeded)
    torch.manual_seed(100 + rank)
    live = nn.Sequential(nn.Linear(4, 3), nn.BatchNorm1d(3))
    dead = nn.Linear(5, 5)                                       # never used: grads stay None (the dead Transformer)
    dp.broadcast_parameters([live, dead])
    flat = torch.cat([p.detach().reshape(-1) for p in list(live.parameters()) + list(dead.parameters())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    # gradient averaging == gradient of the global-batch loss (sum-reduced loss / world)
    x = torch.arange(8 * 4).float().view(8, 4) / 10.0
    xs = x[rank * 4:(rank + 1) * 4]
    red = dp.GradientAllReducer([live, dead])
    live[0](xs).pow(2).sum().backward()
    nbytes = red.reduce()
    assert nbytes == sum(p.numel() for p in live[0].parameters()) * 4       # BN affine got no grad here, dead skipped
    ref = nn.Linear(4, 3)
    ref.load_state_dict(live[0].state_dict())
    (ref(x).pow(2).sum() / world).backward()
    assert torch.allclose(live[0].weight.grad, ref.weight.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(live[0].bias.grad, ref.bias.grad, rtol=1e-5, atol=1e-6)
    assert all(p.grad is None for p in dead.parameters())
    # second step: gradients travel from the hooks, bucket by bucket, during backward
    assert red.buckets and red._armed
    live.zero_grad(set_to_none=True)
    live[0](xs).pow(2).sum().backward()
    assert all(wk is not None for wk in red._work)                            # every bucket was started by a hook
    red.reduce()
    assert torch.allclose(live[0].weight.grad, ref.weight.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(live[0].bias.grad, ref.bias.grad, rtol=1e-5, atol=1e-6)
    # several buckets (tiny cap), reverse registration order, same numbers as the flat call; a gradient that stops
    # arriving is an error, not a silent stale slice
    torch.manual_seed(7)
    net = nn.Sequential(nn.Linear(4, 6), nn.ReLU(), nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 2))
    dp.broadcast_parameters([net])
    net_flat = nn.Sequential(nn.Linear(4, 6), nn.ReLU(), nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 2))
    net_flat.load_state_dict(net.state_dict())
    r_b, r_f = dp.GradientAllReducer([net], bucket_bytes=64), dp.GradientAllReducer([net_flat], overlap=False)
    for step in range(3):
        for m, r in ((net, r_b), (net_flat, r_f)):
            m.zero_grad(set_to_none=True)
            m(xs * (step + 1)).pow(2).sum().backward()
            r.reduce()
        for pa, pb in zip(net.parameters(), net_flat.parameters()):
            assert torch.equal(pa.grad, pb.grad), step
    assert len(r_b.buckets) >= 3 and r_b.buckets[0][2][0] == len(r_b.params) - 1 and len(r_f.buckets) == 1
    # gradient accumulation -- two backward passes per reduce() -- with no_sync() (hooks quiet on the first) and without
    # (the buckets of the first backward are already on the wire: reduce() redoes the exchange): both equal ONE exchange
    # of the accumulated gradients
    for use_ctx in (True, False):
        for m, r in ((net, r_b), (net_flat, r_f)):
            m.zero_grad(set_to_none=True)
            if use_ctx and r is r_b:
                with r.no_sync():
                    m(xs).pow(2).sum().backward()
            else:
                m(xs).pow(2).sum().backward()
            m(xs * 2).pow(2).sum().backward()
            r.reduce()
        for pa, pb in zip(net.parameters(), net_flat.parameters()):
            assert torch.allclose(pa.grad, pb.grad, rtol=1e-6, atol=1e-7), use_ctx
    net.zero_grad(set_to_none=True)
    net[0](xs).sum().backward()                                               # the last two layers get no gradient
    try:
        r_b.reduce()
        raise AssertionError("a missing gradient went unnoticed")
    except RuntimeError as e:
        assert "did not arrive" in str(e)
    r_b.remove_hooks()
    assert dp.all_reduce_max(float(rank), torch.device("cpu")) == float(world - 1)
    dp.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.timeout(180)
def test_dp_helpers_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def _spin_worker(rank, world, port, q):
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "so-net_amd"))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    from sonet_hip import dp
    dp.init_distributed(backend="gloo")
    calls = [0]

    def step():                                                   # a step with a collective in it; rank 1 is four times slower
        time.sleep(0.002 if rank == 0 else 0.008)
        t = torch.ones(1)
        dist.all_reduce(t)
        calls[0] += 1
    n = bench._spin_up(step, 0.3, chunk=4, dev=torch.device("cpu"), sync=lambda: None)
    assert n == calls[0] and n >= 4
    counts = [None] * world
    dist.all_gather_object(counts, n)
    assert counts[0] == counts[1], counts                        # the same number of calls on every rank (or the job hangs)
    dp.barrier()
    dist.destroy_process_group()
    q.put((rank, n))


@pytest.mark.timeout(180)
def test_bench_spin_up_makes_the_same_number_of_calls_on_every_rank():
    """bench.py's untimed spin-up runs for a wall-clock time, and in the training regions its step contains the gradient all-reduce: ranks
    with different clocks must still agree on the number of calls."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_spin_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got[0][1] == got[1][1]


def test_single_process_is_a_no_op():
    import torch.nn as nn
    from sonet_hip import dp
    assert dp.world_size() == 1 and dp.shard_range(5, 1, 0) == (0, 5)
    m = nn.Linear(2, 2)
    m(torch.ones(1, 2)).sum().backward()
    g = m.weight.grad.clone()
    assert dp.GradientAllReducer([m]).reduce() == 0 and torch.equal(m.weight.grad, g)
    dp.broadcast_parameters([m])
    dp.barrier()
