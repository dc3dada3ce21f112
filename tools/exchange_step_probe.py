"""Host time (issue only) and wall time per Chamfer step with and without the exchange, one-rank RCCL group."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, ".")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from pytorch_points_amd import synthetic as S
from pytorch_points_amd.sharded import PackedShardGather
from pytorch_points_amd.network.model_loss import nndistance
dev = torch.device("cuda:0")
B, N = 32, 16384
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev).requires_grad_(True)
x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev).requires_grad_(True)
g1 = torch.full((B, N), 1.0 / (B * N), device=dev); g2 = g1.clone()
ex = PackedShardGather(B, N, N, dev)
pending = []
def plain():
    x1.grad = None; x2.grad = None
    d1, d2, i1, i2 = nndistance(x1, x2)
    torch.autograd.backward([d1, d2], [g1, g2])
def with_ex():
    x1.grad = None; x2.grad = None
    while pending: ex.wait_views(pending.pop())
    d1, d2, i1, i2, h = ex.forward(x1, x2)
    pending.append(h)
    torch.autograd.backward([d1, d2], [g1, g2])
def fwd_only_ex():
    while pending: ex.wait_views(pending.pop())
    with torch.no_grad():
        d1, d2, i1, i2, h = ex.forward(x1, x2)
    pending.append(h)
def measure(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / n
    return host * 1e6, wall * 1e6
for name, fn in (("plain step", plain), ("step with exchange", with_ex), ("forward with exchange", fwd_only_ex)):
    h, w = measure(fn)
    print("%-24s host %.1f us  wall %.1f us" % (name, h, w))
ex.drain()
dist.destroy_process_group()
