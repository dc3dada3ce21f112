import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
from pytorch_points_amd import synthetic as S
from pytorch_points_amd.network.model_loss import nndistance
dev = torch.device("cuda:0")
B, N = 32, 16384
x1 = torch.from_numpy(S.unit_sphere(0, B, N)).to(dev).requires_grad_(True)
x2 = torch.from_numpy(S.unit_sphere(1, B, N)).to(dev).requires_grad_(True)
g1 = torch.full((B, N), 1.0 / (B * N), device=dev); g2 = g1.clone()
def step():
    d1, d2, i1, i2 = nndistance(x1, x2)
    torch.autograd.backward([d1, d2], [g1, g2])
    x1.grad = None; x2.grad = None
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("issue %.1f us/step, complete %.1f us/step" % (t_issue / 500 * 1e6, t_all / 500 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
