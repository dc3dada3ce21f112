"""three_nn over batch sizes (is the query kernel throughput- or latency-bound?)"""
import sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
dev = torch.device("cuda:0")
N, M = 16384, 4096
for B in (4, 8, 16, 32, 64, 128):
    unknown = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); known = torch.from_numpy(S.unit_sphere(1, B, M)).to(dev)
    d2 = torch.empty(B, N, 3, device=dev); idx = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
    for _ in range(3): sampling.three_nn_wrapper(B, N, M, unknown, known, d2, idx)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): sampling.three_nn_wrapper(B, N, M, unknown, known, d2, idx)
    b.record(); torch.cuda.synchronize()
    print("B=%3d  %.4f ms" % (B, a.elapsed_time(b) / 10))
