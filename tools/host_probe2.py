import sys, time, torch
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
def run(tag):
    for _ in range(50): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000): step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: issue %.1f us/step, complete %.1f us/step" % (tag, t_issue / 1000 * 1e6, t_all / 1000 * 1e6))
run("default engine threads")
torch.autograd.set_multithreading_enabled(False)
run("multithreading disabled")
# host-only cost: time issue with a tiny problem (GPU never the bottleneck)
xs1 = x1[:1, :256].detach().clone().requires_grad_(True); xs2 = x2[:1, :256].detach().clone().requires_grad_(True)
gs1 = torch.ones(1, 256, device=dev); gs2 = gs1.clone()
def small():
    d1, d2, i1, i2 = nndistance(xs1, xs2)
    torch.autograd.backward([d1, d2], [gs1, gs2])
    xs1.grad = None; xs2.grad = None
for mt in (True, False):
    torch.autograd.set_multithreading_enabled(mt)
    for _ in range(50): small()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): small()
    t = time.perf_counter() - t0; torch.cuda.synchronize()
    print("host cost per step (tiny problem), engine threads %s: %.1f us" % (mt, t / 2000 * 1e6))
