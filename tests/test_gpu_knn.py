"""GPU parity of knn_points (scan and grid kernels) against the CPU oracle's brute force: indices and
squared distances bit-exact, ties to the lower index, padding conventions, gradients vs autograd of a
torch restatement."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from pytorch_points_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["grid", "scan"])
def knn_path(request, cuda):
    from pytorch_points_amd import _lib
    search = _lib.lib().pp_debug_set_knn_search
    search.argtypes = [ctypes.c_int]
    search.restype = None
    search(0 if request.param == "grid" else 1)
    yield request.param
    search(0)


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


@pytest.mark.parametrize("K", [1, 3, 4, 8, 9, 16, 17, 32])
@pytest.mark.parametrize("b,n,m", [(2, 2048, 2048), (1, 5000, 3001), (3, 1024, 4096)])
def test_knn_matches_oracle(cuda, knn_path, b, n, m, K):
    from pytorch_points_amd.ops import knn_points
    p1 = S.unit_sphere(90, b, n)
    p2 = S.unit_sphere(91, b, m)
    p2[:, m // 2:m // 2 + 40] = p2[:, :40]            # exact ties
    p1[:, :5] = p2[:, 7:12]                           # zero distances
    out = knn_points(_t(p1, cuda), _t(p2, cuda), K=K)
    e_d, e_i = oracle.knn(p1, p2, K)
    assert out.idx.dtype == torch.int64 and out.knn is None
    assert np.array_equal(out.idx.cpu().numpy(), e_i) and np.array_equal(out.dists.cpu().numpy(), e_d)


def test_knn_small_and_ragged(cuda):
    from pytorch_points_amd.ops import knn_points
    p1 = S.unit_sphere(92, 3, 70)
    p2 = S.unit_sphere(93, 3, 50)
    l1 = np.array([70, 13, 0], np.int32)
    l2 = np.array([50, 3, 20], np.int32)
    out = knn_points(_t(p1, cuda), _t(p2, cuda), lengths1=torch.from_numpy(l1), lengths2=torch.from_numpy(l2), K=6,
                     return_nn=True)
    e_d, e_i = oracle.knn(p1, p2, 6, l1, l2)
    assert np.array_equal(out.idx.cpu().numpy(), e_i) and np.array_equal(out.dists.cpu().numpy(), e_d)
    assert out.knn.shape == (3, 70, 6, 3)
    assert torch.equal(out.knn[0, 5, 2], _t(p2, cuda)[0, int(e_i[0, 5, 2])])
    assert float(out.knn[1, :, 3:].abs().max()) == 0.0            # beyond lengths2: zeros
    # K larger than the cloud
    out = knn_points(_t(p1[:, :, :], cuda), _t(p2[:, :4], cuda), K=8)
    e_d, e_i = oracle.knn(p1, p2[:, :4], 8)
    assert np.array_equal(out.idx.cpu().numpy(), e_i) and np.array_equal(out.dists.cpu().numpy(), e_d)


def test_knn_adversarial(cuda, knn_path):
    """offset clouds, planar / collinear / clustered / identical references, far queries"""
    from pytorch_points_amd.ops import knn_points
    n, m, K = 3000, 2048, 8
    cases = []
    cases.append((S.unit_sphere(64, 1, n) + np.float32(1000.0), S.unit_sphere(65, 1, m) + np.float32(1000.0)))
    pl = S.uniform01(66, (1, m, 3)).reshape(1, m, 3).astype(np.float32); pl[..., 1] = -3.0
    cases.append((S.unit_sphere(67, 1, n), pl))
    ln = np.zeros((1, m, 3), np.float32); ln[0, :, 2] = np.linspace(-1, 1, m, dtype=np.float32)
    cases.append((S.unit_sphere(68, 1, n), ln))
    cases.append((S.unit_sphere(69, 1, n) * np.float32(50), S.unit_sphere(70, 1, m)))
    cl = (S.normal(71, (1, m, 3)) * 1e-3).astype(np.float32); cl[0, ::3] += np.float32(0.7)
    cases.append((S.unit_sphere(72, 1, n), cl))
    same = np.concatenate([np.full((1, m, 3), 0.25, np.float32), S.unit_sphere(73, 1, m)], 0)
    cases.append((S.unit_sphere(74, 2, n), same))
    for i, (p1, p2) in enumerate(cases):
        out = knn_points(_t(p1, cuda), _t(p2, cuda), K=K)
        e_d, e_i = oracle.knn(p1, p2, K)
        bad = np.argwhere((out.idx.cpu().numpy() != e_i).any(-1))
        assert bad.size == 0, "case %d: %d rows differ, first %s got %s want %s" % (
            i, len(bad), bad[0], out.idx.cpu().numpy()[tuple(bad[0])], e_i[tuple(bad[0])])
        assert np.array_equal(out.dists.cpu().numpy(), e_d), "case %d distances" % i


def test_knn_gradients(cuda):
    from pytorch_points_amd.ops import knn_points
    b, n, m, K = 2, 300, 200, 5
    p1 = _t(S.unit_sphere(94, b, n), cuda).requires_grad_(True)
    p2 = _t(S.unit_sphere(95, b, m), cuda).requires_grad_(True)
    w = _t(S.normal(96, (b, n, K)), cuda)
    out = knn_points(p1, p2, K=K)
    (out.dists * w).sum().backward()
    g1, g2 = p1.grad.clone(), p2.grad.clone()
    p1.grad = p2.grad = None
    nb = torch.gather(p2.unsqueeze(1).expand(-1, n, -1, -1), 2, out.idx.unsqueeze(-1).expand(-1, -1, -1, 3))
    ref = ((p1.unsqueeze(2) - nb) ** 2).sum(-1)
    (ref * w).sum().backward()
    assert torch.allclose(out.dists, ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(g1, p1.grad, rtol=1e-5, atol=1e-6) and torch.allclose(g2, p2.grad, rtol=1e-5, atol=1e-6)


def test_knn_golden_fixture(cuda, knn_path):
    import os
    from pytorch_points_amd.ops import knn_points
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "knn_b2_n600_m500_k8.npz")))
    out = knn_points(_t(g["p1"], cuda), _t(g["p2"], cuda), K=int(g["K"]))
    assert np.array_equal(out.idx.cpu().numpy(), g["idx"]) and np.array_equal(out.dists.cpu().numpy(), g["dist2"])


@pytest.mark.parametrize("b,n,m,dim,K", [(2, 1024, 1024, 24, 17), (1, 2048, 2048, 64, 17), (2, 700, 1300, 128, 17),
                                         (1, 512, 600, 5, 9), (1, 300, 257, 130, 33), (2, 256, 512, 3, 48),
                                         (1, 200, 1000, 16, 128), (1, 64, 40, 7, 64)])
def test_knn_any_dimension_and_large_k(cuda, b, n, m, dim, K):
    """Feature-space searches (reference network/layers.py:52,99: D = channel count, K = k + 1 = 17) and K > 32:
    indices and squared distances bit-exact against the generalised oracle (sequential fma chain over D)."""
    from pytorch_points_amd.ops import knn_points
    p1 = S.normal(95, (b, n, dim))
    p2 = S.normal(96, (b, m, dim))
    p2[:, m // 2:m // 2 + 20] = p2[:, :20]            # exact ties: the lower index first
    p1[:, :3] = p2[:, 4:7]                            # zero distances
    out = knn_points(_t(p1, cuda), _t(p2, cuda), K=K, return_nn=True)
    e_d, e_i = oracle.knn(p1, p2, K)
    assert np.array_equal(out.idx.cpu().numpy(), e_i) and np.array_equal(out.dists.cpu().numpy(), e_d)
    assert out.knn.shape == (b, n, K, dim)
    if K <= m:
        assert torch.equal(out.knn[0, 10, K - 1], _t(p2, cuda)[0, int(e_i[0, 10, K - 1])])


def test_knn_feature_space_gradients_and_ragged(cuda):
    from pytorch_points_amd.ops import knn_points
    b, n, m, dim, K = 2, 200, 150, 12, 5
    p1 = _t(S.normal(97, (b, n, dim)), cuda).requires_grad_(True)
    p2 = _t(S.normal(98, (b, m, dim)), cuda).requires_grad_(True)
    l2 = torch.tensor([150, 40])
    out = knn_points(p1, p2, lengths2=l2, K=K)
    e_d, e_i = oracle.knn(p1.detach().cpu().numpy(), p2.detach().cpu().numpy(), K, None, l2.numpy().astype(np.int32))
    assert np.array_equal(out.idx.cpu().numpy(), e_i) and np.array_equal(out.dists.detach().cpu().numpy(), e_d)
    w = _t(S.normal(99, (b, n, K)), cuda)
    (out.dists * w).sum().backward()
    q1 = p1.detach().double().requires_grad_(True)
    q2 = p2.detach().double().requires_grad_(True)
    nb = torch.gather(q2.unsqueeze(1).expand(-1, n, -1, -1), 2, out.idx.unsqueeze(-1).expand(-1, -1, -1, dim))
    (((q1.unsqueeze(2) - nb) ** 2).sum(-1) * w.double()).sum().backward()
    assert torch.allclose(p1.grad.double(), q1.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(p2.grad.double(), q2.grad, rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError):
        knn_points(p1, p2, K=129)
