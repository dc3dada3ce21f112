"""world_size-2 gloo test of the batch-shard + all-gather host logic (CPU, no GPU).

The local operator is injected (a torch restatement of Chamfer) because the product operator is
HIP-only; what is under test is pytorch_points_amd/sharded.py: slab bounds, gather order, ragged
shards, and that gradients reach exactly the local shard."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorch_points_amd import sharded


def _torch_chamfer(x1, x2):
    d = ((x1[:, :, None] - x2[:, None]) ** 2).sum(-1)
    d1, i1 = d.min(2)
    d2, i2 = d.min(1)
    return d1, d2, i1.int(), i2.int()


def _worker(rank, world, port, sizes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = sum(sizes)
        g = torch.Generator().manual_seed(0)
        X1 = torch.randn(B, 40, 3, generator=g)
        X2 = torch.randn(B, 30, 3, generator=g)
        lo = sum(sizes[:rank])
        hi = lo + sizes[rank]
        x1 = X1[lo:hi].clone().requires_grad_(True)
        x2 = X2[lo:hi].clone().requires_grad_(True)
        d1, d2, i1, i2 = sharded.sharded_nndistance(x1, x2, _local_op=_torch_chamfer)
        F1 = X1.clone().requires_grad_(True)
        F2 = X2.clone().requires_grad_(True)
        e1, e2, j1, j2 = _torch_chamfer(F1, F2)
        ok = torch.equal(d1, e1) and torch.equal(d2, e2) and torch.equal(i1, j1) and torch.equal(i2, j2)
        ok = ok and d1.shape[0] == B and not i1.requires_grad
        w = torch.arange(B, dtype=torch.float32)[:, None]
        ((d1 * w).sum() + d2.sum()).backward()
        ((e1 * w).sum() + e2.sum()).backward()
        ok = ok and torch.allclose(x1.grad, F1.grad[lo:hi]) and torch.allclose(x2.grad, F2.grad[lo:hi])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("sizes", [[3, 3], [4, 1]])
def test_sharded_equals_unsharded_world2(sizes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_bounds_partition_the_batch():
    for B in (0, 1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [sharded.shard_bounds(B, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_single_process_is_identity():
    x = torch.randn(2, 3)
    assert sharded.all_gather_batch(x) is x


def _packed_worker(rank, world, port, n, m, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 3
        ex = sharded.PackedShardGather(B, n, m, torch.device("cpu"))
        ok = True
        for step in range(3):   # more steps than buffers: slots are reused
            g = torch.Generator().manual_seed(100 + step)
            D1 = torch.rand(world * B, n, generator=g)
            D2 = torch.rand(world * B, m, generator=g)
            I1 = torch.randint(0, m, (world * B, n), generator=g, dtype=torch.int32)
            I2 = torch.randint(0, n, (world * B, m), generator=g, dtype=torch.int32)
            I1[:, 0] = m - 1        # the largest index (65534 in the compact case: above int16's range)
            I2[:, -1] = n - 1
            I1[:, 1] = -1           # labeled Chamfer's "no partner": must survive the 16-bit packing
            lo, hi = rank * B, (rank + 1) * B
            if step == 1:   # the in-place form: the producer writes its distances into the slot's own fields
                slot, v1, v2 = ex.begin()
                v1.copy_(D1[lo:hi]); v2.copy_(D2[lo:hi])
                h = ex.launch_in_place(slot, I1[lo:hi], I2[lo:hi])
            else:
                h = ex.launch(D1[lo:hi], D2[lo:hi], I1[lo:hi], I2[lo:hi])
            # a collective of the caller's own between launch and wait (the exchange is issued in program order)
            t = torch.ones(3) * (rank + 1)
            dist.all_reduce(t)
            ok = ok and float(t[0]) == sum(range(1, world + 1))
            d1, d2, i1, i2 = ex.wait(h)
            ok = ok and torch.equal(d1, D1) and torch.equal(d2, D2) and torch.equal(i1, I1) and torch.equal(i2, I2)
            ok = ok and i1.dtype == torch.int32
            # ... and as views of the gathered buffer: (world, B, n | m), indices as the words they travelled as
            w1, w2, j1, j2 = ex.wait_views(h)
            ok = ok and w1.shape == (world, B, n) and torch.equal(w1.reshape(world * B, n), D1)
            ok = ok and torch.equal(w2.reshape(world * B, m), D2)
            mask = 0xFFFF if ex.compact else -1
            ok = ok and torch.equal(j1.to(torch.int32).reshape(world * B, n) & mask, I1 & mask)
            ok = ok and torch.equal(j2.to(torch.int32).reshape(world * B, m) & mask, I2 & mask)
        ex.drain()
        q.put((rank, bool(ok), ex.compact))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,m,compact", [(50, 65535, True), (50, 65536, False), (70000, 33, False)])
def test_packed_shard_gather_world2(n, m, compact):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, n, m, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True, compact), (1, True, compact)]


def _pipeline_worker(rank, world, port, depth, q, exchange=None):
    """the control flow of the RCCL path (bench.py / PackedShardGather.forward): begin -> the search writes its
    distances in place -> launch_in_place -> a collective of the caller's -> wait_views of the step BEFORE (the gather
    of step k overlaps step k + 1), slots reused several times over"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, n, m = 2, 37, 53
        ex = sharded.PackedShardGather(B, n, m, torch.device("cpu"), depth=depth, exchange=exchange)
        ok = exchange is None or (ex.p2p == (exchange == "p2p"))
        pending = []     # (slot, step) launched and not yet read
        steps = 3 * depth + 1

        def data(step):
            g = torch.Generator().manual_seed(1000 + step)
            return (torch.rand(world * B, n, generator=g), torch.rand(world * B, m, generator=g),
                    torch.randint(-1, m, (world * B, n), generator=g, dtype=torch.int32),
                    torch.randint(-1, n, (world * B, m), generator=g, dtype=torch.int32))

        def check(slot, step):
            D1, D2, I1, I2 = data(step)
            w1, w2, j1, j2 = ex.wait_views(slot)
            good = w1.shape == (world, B, n) and torch.equal(w1.reshape(world * B, n), D1)
            good = good and torch.equal(w2.reshape(world * B, m), D2)
            good = good and torch.equal(j1.to(torch.int32).reshape(world * B, n) & 0xFFFF, I1 & 0xFFFF)
            good = good and torch.equal(j2.to(torch.int32).reshape(world * B, m) & 0xFFFF, I2 & 0xFFFF)
            return good

        lo, hi = rank * B, (rank + 1) * B
        for step in range(steps):
            D1, D2, I1, I2 = data(step)
            slot, v1, v2 = ex.begin()
            v1.copy_(D1[lo:hi]); v2.copy_(D2[lo:hi])          # "the search writes its distances into the slot"
            h = ex.launch_in_place(slot, I1[lo:hi], I2[lo:hi])
            ok = ok and h == slot
            t = torch.ones(2) * (rank + 1)                     # the caller's own collective (e.g. a gradient all-reduce)
            dist.all_reduce(t)
            ok = ok and float(t[0]) == sum(range(1, world + 1))
            pending.append((slot, step))
            while len(pending) >= depth:                       # read a step once `depth` - 1 newer ones are in flight
                ok = ok and check(*pending.pop(0))
        while pending:
            ok = ok and check(*pending.pop(0))
        ex.drain()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", [None, "p2p"])
@pytest.mark.parametrize("world,depth", [(2, 2), (2, 3), (3, 2), (3, 3)])
def test_packed_shard_gather_pipelined(world, depth, exchange):
    """exchange="p2p": the all-gather as world - 1 sends of the own row and world - 1 receives into the other rows,
    in place (the control flow of PP_SHARD_EXCHANGE=p2p over RCCL: one grouped set per step)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, depth, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]
