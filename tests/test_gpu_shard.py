"""pack / unpack kernels of the batch-sharded exchange (csrc/shard.hip) against the tensor-op form of
pytorch_points_amd/sharded.py, for a fabricated world of three ranks (single process, no collective)."""
import numpy as np
import pytest
import torch

from pytorch_points_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,n,m", [(3, 1000, 777), (2, 65535, 40), (2, 65536, 40), (1, 5, 70000), (4, 16384, 16384)])
def test_shard_pack_unpack_roundtrip(cuda, b, n, m):
    world = 3
    compact = max(n, m) <= 65535
    g = torch.Generator(device="cpu").manual_seed(b * 131 + n)
    D1 = torch.rand(world * b, n, generator=g).to(cuda)
    D2 = torch.rand(world * b, m, generator=g).to(cuda)
    I1 = torch.randint(0, m, (world * b, n), generator=g, dtype=torch.int32).to(cuda)
    I2 = torch.randint(0, n, (world * b, m), generator=g, dtype=torch.int32).to(cuda)
    I1[:, 0] = m - 1
    I2[:, -1] = n - 1
    I1[:, 1] = -1      # labeled Chamfer's "no partner" (0xFFFF in the 16-bit form)
    L = _lib.lib()
    stride = int(L.pp_shard_packed_bytes(b * n, b * m, 1 if compact else 0))
    assert stride % 16 == 0 and stride >= (b * n + b * m) * (6 if compact else 8)
    recv = torch.zeros(world, stride, dtype=torch.uint8, device=cuda)
    with _lib.on_device(cuda) as stream:
        for r in range(world):
            sl = slice(r * b, (r + 1) * b)
            _lib.check(L.pp_shard_pack_f32(_lib.ptr(D1[sl].contiguous()), _lib.ptr(D2[sl].contiguous()),
                                           _lib.ptr(I1[sl].contiguous()), _lib.ptr(I2[sl].contiguous()),
                                           _lib.ptr(recv[r]), b * n, b * m, 1 if compact else 0, stream), "pack")
        o1 = torch.empty_like(D1); o2 = torch.empty_like(D2); j1 = torch.empty_like(I1); j2 = torch.empty_like(I2)
        _lib.check(L.pp_shard_unpack_f32(_lib.ptr(recv), world, stride, b * n, b * m, 1 if compact else 0,
                                         _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(j1), _lib.ptr(j2), stream), "unpack")
    assert torch.equal(o1, D1) and torch.equal(o2, D2) and torch.equal(j1, I1) and torch.equal(j2, I2)
    # round 4: distances IN PLACE (the search wrote them into the packed buffer itself: dist pointers NULL) and an
    # indices-only unpack (the distances are read where they were gathered)
    recv2 = torch.zeros(world, stride, dtype=torch.uint8, device=cuda)
    with _lib.on_device(cuda) as stream:
        for r in range(world):
            sl = slice(r * b, (r + 1) * b)
            fl = recv2[r][: 4 * b * (n + m)].view(torch.float32)
            fl[: b * n] = D1[sl].reshape(-1)
            fl[b * n:] = D2[sl].reshape(-1)
            _lib.check(L.pp_shard_pack_f32(None, None, _lib.ptr(I1[sl].contiguous()), _lib.ptr(I2[sl].contiguous()),
                                           _lib.ptr(recv2[r]), b * n, b * m, 1 if compact else 0, stream), "pack")
        k1 = torch.empty_like(I1); k2 = torch.empty_like(I2)
        _lib.check(L.pp_shard_unpack_f32(_lib.ptr(recv2), world, stride, b * n, b * m, 1 if compact else 0,
                                         None, None, _lib.ptr(k1), _lib.ptr(k2), stream), "unpack")
    assert torch.equal(recv2[:, : (6 if compact else 8) * b * (n + m)], recv[:, : (6 if compact else 8) * b * (n + m)])
    assert torch.equal(k1, I1) and torch.equal(k2, I2)
    # the packed bytes are what the tensor-op form (CPU / gloo path) produces
    row = recv[1].cpu()
    f = row[: 4 * b * (n + m)].view(torch.float32)
    assert torch.equal(f[: b * n], D1[b:2 * b].reshape(-1).cpu()) and torch.equal(f[b * n:], D2[b:2 * b].reshape(-1).cpu())
    if compact:
        i = row[4 * b * (n + m): 6 * b * (n + m)].view(torch.int16).to(torch.int32) & 0xFFFF
        i = torch.where(i == 0xFFFF, torch.full_like(i, -1), i)
    else:
        i = row[4 * b * (n + m): 8 * b * (n + m)].view(torch.int32)
    assert torch.equal(i[: b * n], I1[b:2 * b].reshape(-1).cpu()) and torch.equal(i[b * n:], I2[b:2 * b].reshape(-1).cpu())


_ONE_RANK_EXCHANGE = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pytorch_points_amd.sharded import PackedShardGather
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda:0")
B, N, M = 3, 5000, 70000                                  # M > 65536: 32-bit indices; then a 16-bit case
for (n, m) in ((N, M), (4096, 2048)):
    ex = PackedShardGather(B, n, m, dev)
    mode = os.environ.get('PP_SHARD_EXCHANGE', 'native')
    assert bool(getattr(ex, 'direct', False)) == (mode in ('rccl', 'rccl_p2p')), 'exchange path'
    assert bool(ex.p2p) == (mode in ('p2p', 'rccl_p2p', 'python_p2p')), 'exchange path (p2p)'
    fail = os.environ.get('PP_SHARD_SELFCHECK_FAIL', '0') == '1'
    gen = torch.Generator(device="cpu").manual_seed(n)
    steps = []
    for s in range(5):                                     # more launches than slots: the slots are reused
        d1 = torch.rand(B, n, generator=gen).to(dev); d2 = torch.rand(B, m, generator=gen).to(dev)
        i1 = torch.randint(0, m, (B, n), generator=gen, dtype=torch.int32).to(dev)
        i2 = torch.randint(0, n, (B, m), generator=gen, dtype=torch.int32).to(dev)
        h = ex.launch(d1, d2, i1, i2)
        # something else on the launch stream while the exchange runs beside it -- and a collective of the caller's own
        # on the same group between launch and wait (ADVICE r3: the exchange is issued in program order)
        torch.empty(1 << 22, device=dev).normal_()
        t = torch.ones(4, device=dev); dist.all_reduce(t)
        g = ex.wait(h)
        if fail:                                           # the first wait found the direct path wanting: c10d from here on
            assert not ex.direct and not ex.p2p and ex._checked
        for a, e in zip(g, (d1, d2, i1, i2)):
            assert a.dtype == e.dtype and torch.equal(a, e), (n, m, s)
        v = ex.wait_views(h)                               # the same, as views of the gathered buffer
        assert v[0].shape == (1, B, n) and v[1].shape == (1, B, m) and v[0].dtype == torch.float32
        assert torch.equal(v[0][0], d1) and torch.equal(v[1][0], d2)
        w1 = v[2][0].to(torch.int32) & (0xFFFF if ex.compact else -1)
        assert torch.equal(w1, i1 & (0xFFFF if ex.compact else -1)), (n, m, s)
    ex.drain()
    # the in-place form: the search writes its distances into the slot, only the indices are narrowed in
    from pytorch_points_amd.network.model_loss import nndistance
    from pytorch_points_amd import synthetic as S
    if n <= 8192:
        x1 = torch.from_numpy(S.unit_sphere(7, B, n)).to(dev).requires_grad_(True)
        x2 = torch.from_numpy(S.unit_sphere(8, B, m)).to(dev).requires_grad_(True)
        r1, r2, j1, j2 = nndistance(x1, x2)
        (r1.mean() + r2.mean()).backward()
        gref = (x1.grad.clone(), x2.grad.clone()); x1.grad = None; x2.grad = None
        for s in range(3):
            e1, e2, k1, k2, h = ex.forward(x1, x2)
            assert torch.equal(e1, r1) and torch.equal(e2, r2) and torch.equal(k1, j1) and torch.equal(k2, j2)
            (e1.mean() + e2.mean()).backward()
            assert torch.equal(x1.grad, gref[0]) and torch.equal(x2.grad, gref[1])
            x1.grad = None; x2.grad = None
            G1, G2, H1, H2 = ex.wait(h)
            assert torch.equal(G1, r1.detach()) and torch.equal(G2, r2.detach()) and torch.equal(H1, j1) and torch.equal(H2, j2)
        ex.drain()
torch.cuda.synchronize()
dist.destroy_process_group()
print("exchange ok")
"""


_EXCHANGE_PATHS = {"default": (None, "29539"), "native": ("native", "29541"), "python": ("python", "29543"),
                   "rccl": ("rccl", "29545"), "rccl_selfcheck_fail": ("rccl", "29547"), "p2p": ("p2p", "29549"),
                   "p2p_selfcheck_fail": ("p2p", "29551"), "rccl_p2p": ("rccl_p2p", "29553"),
                   "python_p2p": ("python_p2p", "29555")}


@pytest.mark.parametrize("path", list(_EXCHANGE_PATHS))
def test_packed_exchange_on_one_rank_rccl_group(cuda, tmp_path, path):
    """PackedShardGather on the GPU path proper: RCCL all-gather (a one-rank group: this box has one GPU),
    slot reuse, 16- and 32-bit indices, the gathered result as contiguous tensors (wait) and as views of the gathered
    buffer (wait_views), the in-place form (forward: the search writes into the slot), a collective of the caller's
    between launch and wait.  native: the exchange as one C++ call (csrc/torch_bridge.cpp: PackedExchange over c10d);
    python: the same steps issued from Python (PP_SHARD_EXCHANGE=python); rccl: the native call with the all-gather as
    a direct ncclAllGather on the exchange object's own communicator (opt-in), its first exchange checked against c10d;
    rccl_selfcheck_fail: that check made to fail (PP_SHARD_SELFCHECK_FAIL=1) -- every slot is gathered again over c10d and
    the results are still the shards'; p2p (round 6): the all-gather as one grouped set of sends and receives between
    the rows, in place (on one rank: nothing moves, the own row is where the search wrote it), over c10d's communicator,
    over the object's own (rccl_p2p) or issued from Python (python_p2p), and its self-check made to fail; default: no
    variable set (= native).  In a subprocess: the process group must not
    leak into the other tests."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "one_rank_exchange.py"
    script.write_text(_ONE_RANK_EXCHANGE)
    env = dict(os.environ)
    env.pop("PP_SHARD_EXCHANGE", None)
    env.pop("PP_SHARD_SELFCHECK_FAIL", None)
    mode, port = _EXCHANGE_PATHS[path]
    if mode is not None:
        env["PP_SHARD_EXCHANGE"] = mode
    if path.endswith("_selfcheck_fail"):
        env["PP_SHARD_SELFCHECK_FAIL"] = "1"
    out = subprocess.run([sys.executable, str(script), root, port], capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0 and "exchange ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
