"""Host time per call of the pieces of PackedShardGather.launch on a one-rank RCCL group (the exchange is host-bound
at config 2: 0.19 ms per step with it against 0.08 without).  RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, ".")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
from pytorch_points_amd import _lib
from pytorch_points_amd.sharded import PackedShardGather
dev = torch.device("cuda:0")
B, N = 32, 16384
ex = PackedShardGather(B, N, N, dev)
d1 = torch.rand(B, N, device=dev); d2 = torch.rand(B, N, device=dev)
i1 = torch.randint(0, N, (B, N), device=dev, dtype=torch.int32); i2 = i1.clone()
def host(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return dt * 1e6
print("launch + wait (whole exchange)      %.1f us" % host(lambda: ex.wait(ex.launch(d1, d2, i1, i2))))
send, recv = ex.send[0], ex.recv[0]
print("all_gather_into_tensor async        %.1f us" % host(lambda: dist.all_gather_into_tensor(recv, send, async_op=True)))
def pack():
    with _lib.on_device(dev) as stream:
        _lib.lib().pp_shard_pack_f32(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(i1), _lib.ptr(i2), _lib.ptr(send), B * N, B * N, 1, stream)
print("pack (on_device + ctypes)           %.1f us" % host(pack))
side = torch.cuda.Stream()
ev = torch.cuda.Event()
def side_part():
    with torch.cuda.stream(side):
        ev.record(side)
    torch.cuda.current_stream().wait_event(ev)
print("stream ctx + event record + wait    %.1f us" % host(side_part))
w = dist.all_gather_into_tensor(recv, send, async_op=True)
print("work.wait()                         %.1f us" % host(lambda: w.wait()))
dist.destroy_process_group()
