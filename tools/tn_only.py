"""three_nn alone at the interpolation shape (for profiling passes)"""
import sys, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd._ext import sampling
dev = torch.device("cuda:0")
B, N, M = 32, 16384, 4096
unknown = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev); known = torch.from_numpy(S.unit_sphere(1, B, M)).to(dev)
d2 = torch.empty(B, N, 3, device=dev); idx = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
for _ in range(6): sampling.three_nn_wrapper(B, N, M, unknown, known, d2, idx)
torch.cuda.synchronize()
