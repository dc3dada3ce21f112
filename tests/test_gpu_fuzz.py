"""Randomised shapes and point distributions for every search operator (Chamfer, labeled Chamfer,
ball_query, three_nn, knn) against the CPU oracle, bit for bit.  Seeds are fixed: a failure names the
case.  Sizes straddle the thresholds at which the operators switch between scan and grid kernels."""
import os

import numpy as np
import pytest
import torch

import oracle
from pytorch_points_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "grid"])
def search_mode(request, cuda):
    """auto: the operators' own choice between scan and grid kernels; grid: the grid kernels wherever
    they are structurally possible (the automatic choice leaves small problems to the scans)"""
    import ctypes
    from pytorch_points_amd import _lib
    knobs = []
    for name in ("pp_debug_set_nmdistance_search", "pp_debug_set_ball_query_search"):
        f = getattr(_lib.lib(), name)
        f.argtypes = [ctypes.c_int]
        f.restype = None
        f(2 if request.param == "grid" else 0)
        knobs.append(f)
    yield request.param
    for f in knobs:
        f(0)


def _cloud(rng, b, n, kind):
    """a (b, n, 3) fp32 cloud of a random family"""
    if kind == 0:
        x = S.unit_sphere(int(rng.integers(1 << 30)), b, n)
    elif kind == 1:
        x = rng.random((b, n, 3), dtype=np.float32)
    elif kind == 2:   # clusters of very different scales
        c = rng.random((b, 7, 3), dtype=np.float32) * 4
        x = c[:, rng.integers(0, 7, n)] + (rng.standard_normal((b, n, 3)) * rng.choice([1e-3, 1e-2, 0.3])).astype(np.float32)
    elif kind == 3:   # quantised coordinates: many exact ties and duplicates
        x = (rng.integers(0, 12, (b, n, 3)) / 4).astype(np.float32)
    elif kind == 4:   # thin slab far from the origin
        x = rng.random((b, n, 3), dtype=np.float32) * np.array([3, 3, 1e-3], np.float32) + np.float32(300.0)
    elif kind == 5:   # one far outlier stretches the bounding box
        x = rng.random((b, n, 3), dtype=np.float32)
        x[:, 0] = 1e3
    elif kind == 6:   # Gaussian cloud: dense core, thin tails (far queries, group search)
        x = rng.standard_normal((b, n, 3)).astype(np.float32) * np.float32(rng.choice([0.1, 1.0, 30.0]))
    else:             # two scales: a share of the points in a tiny corner of the box (crowded cells, second-level grids)
        x = rng.random((b, n, 3), dtype=np.float32)
        k = int(n * rng.choice([0.1, 0.5, 0.9]))
        x[:, :k] *= np.float32(rng.choice([1e-2, 1e-3]))
    return np.ascontiguousarray(x, np.float32)


def _sizes(rng):
    b = int(rng.integers(1, 5))
    n = int(rng.choice([1, 7, 64, 300, 1023, 1024, 2047, 2048, 2500, 4096, 5003, 8192, 16384]))
    m = int(rng.choice([1, 5, 64, 333, 1024, 2047, 2048, 3000, 4096, 6001, 9000, 16384]))
    return b, n, m


@pytest.mark.parametrize("seed", range(int(os.environ.get("PP_FUZZ_SEEDS", "100"))))   # PP_FUZZ_SEEDS=1000 for a long run
def test_fuzz_chamfer_and_labeled(cuda, search_mode, seed):
    from pytorch_points_amd.network.model_loss import nndistance, labeled_nndistance
    rng = np.random.default_rng(1000 + seed)
    b, n, m = _sizes(rng)
    k1, k2 = int(rng.integers(0, 8)), int(rng.integers(0, 8))
    x1, x2 = _cloud(rng, b, n, k1), _cloud(rng, b, m, k2)
    if seed % 3 == 0:
        x2 = x2 + x1.mean(1, keepdims=True) - x2.mean(1, keepdims=True)   # overlapping clouds
    got = nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda))
    exp = oracle.chamfer_forward(x1, x2)
    for g, e, what in zip((got[0], got[2], got[1], got[3]), exp, ("dist1", "idx1", "dist2", "idx2")):
        assert np.array_equal(g.cpu().numpy(), e), "seed %d (b=%d n=%d m=%d kinds %d/%d): %s" % (seed, b, n, m, k1, k2, what)
    nl = int(rng.choice([1, 2, 5, 400]))
    l1 = rng.integers(0, nl, (b, n)).astype(np.float32)
    l2 = rng.integers(0, max(1, nl - (seed % 2)), (b, m)).astype(np.float32)
    got = labeled_nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda),
                             torch.from_numpy(l1).to(cuda), torch.from_numpy(l2).to(cuda))
    exp = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    for g, e, what in zip((got[0], got[2], got[1], got[3]), exp, ("dist1", "idx1", "dist2", "idx2")):
        assert np.array_equal(g.cpu().numpy(), e), "labeled seed %d (b=%d n=%d m=%d nl=%d): %s" % (seed, b, n, m, nl, what)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_ball_query(cuda, search_mode, seed):
    from pytorch_points_amd._ext import sampling
    rng = np.random.default_rng(2000 + seed)
    b, m, n = _sizes(rng)                      # m centres, n points
    kind = int(rng.integers(0, 8))
    x = _cloud(rng, b, n, kind)
    c = x[:, rng.integers(0, n, m)] if seed % 2 else _cloud(rng, b, m, int(rng.integers(0, 8)))
    c = np.ascontiguousarray(c + (rng.standard_normal(c.shape) * 1e-3).astype(np.float32))
    ext = float(np.ptp(x.reshape(-1, 3), 0).max()) or 1.0
    r = float(rng.choice([1e-4, 0.01, 0.05, 0.2, 1.5])) * min(ext, 3.0)
    ns = int(rng.choice([1, 3, 16, 33, 64, 128]))
    got = sampling.ball_query(torch.from_numpy(c).to(cuda), torch.from_numpy(x).to(cuda), r, ns)
    exp = oracle.ball_query(c, x, r, ns)
    assert np.array_equal(got.cpu().numpy(), exp), "seed %d: b=%d n=%d m=%d kind %d r=%g ns=%d" % (seed, b, n, m, kind, r, ns)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_three_nn_and_knn(cuda, seed):
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.ops import knn_points
    rng = np.random.default_rng(3000 + seed)
    b, n, m = _sizes(rng)
    u, k = _cloud(rng, b, n, int(rng.integers(0, 8))), _cloud(rng, b, m, int(rng.integers(0, 8)))
    d2 = torch.empty(b, n, 3, device=cuda)
    idx = torch.empty(b, n, 3, dtype=torch.int32, device=cuda)
    sampling.three_nn_wrapper(b, n, m, torch.from_numpy(u).to(cuda), torch.from_numpy(k).to(cuda), d2, idx)
    e_d, e_i = oracle.three_nn(u, k)
    assert np.array_equal(idx.cpu().numpy(), e_i) and np.array_equal(d2.cpu().numpy(), e_d), \
        "three_nn seed %d b=%d n=%d m=%d" % (seed, b, n, m)
    K = int(rng.choice([1, 2, 5, 8, 13, 16, 32]))
    out = knn_points(torch.from_numpy(u).to(cuda), torch.from_numpy(k).to(cuda), K=K)
    e_d, e_i = oracle.knn(u, k, K)
    assert np.array_equal(out.idx.cpu().numpy(), e_i) and np.array_equal(out.dists.cpu().numpy(), e_d), \
        "knn seed %d b=%d n=%d m=%d K=%d" % (seed, b, n, m, K)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PP_FUZZ_SEEDS", "100")) // 2))
def test_fuzz_far_and_offset_clouds(cuda, seed):
    """round 3: clouds at random offsets and scales from each other (far-field group search: finite rim cells of an
    untrimmed box, member-by-member row cuts, sifted candidates), with and without a few outliers (trimmed boxes),
    far from the origin or not -- grid search forced, against the oracle"""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd.network.model_loss import nndistance
    rng = np.random.default_rng(7000 + seed)
    b = int(rng.integers(1, 3))
    n = int(rng.choice([2048, 3000, 4096, 6001, 8192]))
    m = int(rng.choice([2048, 2500, 4096, 7000, 8192]))
    x1 = _cloud(rng, b, n, int(rng.integers(0, 8)))
    x2 = _cloud(rng, b, m, int(rng.integers(0, 8)))
    x2 = x2 * np.float32(rng.choice([0.01, 0.3, 1.0, 5.0]))
    direction = rng.standard_normal(3).astype(np.float32)
    direction /= np.linalg.norm(direction)
    x2 = x2 + direction * np.float32(rng.choice([0.0, 0.7, 3.0, 40.0]))
    if seed % 3 == 0:      # a few outliers of one cloud inside / beside the other (a trimmed box with a populated rim)
        k = int(rng.integers(1, 6))
        x2[:, :k] = x1[:, :k] + np.float32(1e-3)
    if seed % 5 == 0:      # both far from the origin
        shift = np.float32(rng.choice([100.0, 3000.0]))
        x1, x2 = x1 + shift, x2 + shift
    x1, x2 = np.ascontiguousarray(x1, np.float32), np.ascontiguousarray(x2, np.float32)
    knob = _lib.lib().pp_debug_set_nmdistance_search
    knob.argtypes = [ctypes.c_int]
    knob.restype = None
    knob(2)
    try:
        got = nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda))
    finally:
        knob(0)
    exp = oracle.chamfer_forward(x1, x2)
    for g, e, what in zip((got[0], got[2], got[1], got[3]), exp, ("dist1", "idx1", "dist2", "idx2")):
        assert np.array_equal(g.cpu().numpy(), e), "seed %d (b=%d n=%d m=%d): %s differs at %d places" % (
            seed, b, n, m, what, int((g.cpu().numpy() != e).sum()))


@pytest.mark.parametrize("seed", range(int(os.environ.get("PP_FUZZ_SEEDS", "100")) // 2))
def test_fuzz_furthest_sampling(cuda, seed):
    """furthest_sampling through the operator's own choice of kernel -- the bucketed kernel from 2048 points and 32
    picks, the cluster / single-block kernels below -- on random shapes, seeds and cloud families (ties and duplicates,
    clusters of very different scales, thin slabs far from the origin, outliers), sometimes with a random incoming
    temp: picks AND temp against the oracle, bit for bit"""
    from pytorch_points_amd._ext import sampling
    rng = np.random.default_rng(7000 + seed)
    b = int(rng.integers(1, 4))
    n = int(rng.choice([300, 1024, 2048, 2049, 3000, 4096, 5003, 8192, 12000, 16384, 20000, 33000, 40000]))
    m = int(rng.choice([1, 5, 31, 32, 33, 64, 200, 513, 1024, 1700]))
    m = min(m, n)
    # the bucketed kernel's chain: the library's choice (several picks per round from 32768 points or 1024 picks), or
    # several picks per round wherever the bucketed kernel runs (round 6)
    import ctypes
    from pytorch_points_amd import _lib
    chain = _lib.lib().pp_debug_set_fps_bucket_chain
    chain.argtypes = [ctypes.c_int]
    chain.restype = None
    force_batched = bool(rng.integers(0, 2))
    x = _cloud(rng, b, n, int(rng.integers(0, 8)))
    start = int(rng.integers(0, n))
    t0 = None
    if rng.random() < 0.3:   # an incoming temp that is not the reference's 1e10 fill
        t0 = (np.abs(rng.standard_normal((b, n))) * float(rng.choice([1e-3, 0.05, 10.0]))).astype(np.float32)
    e_idx, e_temp = oracle.furthest_sampling(x, m, start, temp=t0)
    idx = torch.empty(b, m, dtype=torch.int32, device=cuda)
    temp = torch.from_numpy(t0).to(cuda) if t0 is not None else torch.full((b, n), 1e10, dtype=torch.float32, device=cuda)
    pts = torch.empty(b, m, 3, device=cuda)
    chain(2 if force_batched else 0)
    try:
        sampling.furthest_sampling(m, start, torch.from_numpy(x).to(cuda), temp, idx, pts, False)
    finally:
        chain(0)
    assert np.array_equal(idx.cpu().numpy(), e_idx), (b, n, m, start, force_batched)
    assert np.array_equal(temp.cpu().numpy(), e_temp), (b, n, m, start)
    assert np.array_equal(pts.cpu().numpy(), np.take_along_axis(x, e_idx[..., None].astype(np.int64), 1))


@pytest.mark.parametrize("seed", range(int(os.environ.get("PP_FUZZ_SEEDS", "100")) // 4))
def test_fuzz_config2_class(cuda, seed):
    """the default search on random sizes of config 2's class (8192 .. 17408 points, multiples of four: what the build
    sorts as one chunk) and cloud families -- surfaces (ellipsoids, noisy / double shells, open sheets), volumes, clusters,
    far clouds -- against the every-pair kernel, bit for bit; launched twice on the same workspace"""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd.network.model_loss import nndistance
    rng = np.random.default_rng(9000 + seed)
    b = int(rng.integers(1, 4))
    n = 4 * int(rng.integers(2048, 4353))          # [8192, 17408], multiples of four
    m = 4 * int(rng.integers(2048, 4353))
    def surface(k):
        x = S.unit_sphere(int(rng.integers(1 << 30)), b, k)
        what = int(rng.integers(0, 5))
        if what == 1:     # an ellipsoid: layers of unequal population along z
            x = x * np.array([1.0, 0.6, float(rng.choice([0.3, 1.7]))], np.float32)
        elif what == 2:   # a noisy shell
            x = x * (1 + 0.02 * rng.standard_normal((b, k, 1))).astype(np.float32)
        elif what == 3:   # two shells
            x[:, ::2] *= np.float32(0.55)
        elif what == 4:   # an open sheet: half of the sphere folded onto the other half
            x[..., 2] = np.abs(x[..., 2])
        return np.ascontiguousarray(x, np.float32)
    kind = int(rng.integers(0, 10))
    if kind < 6:
        x1, x2 = surface(n), surface(m)
    else:                 # whatever the other fuzz tests use: mostly declined
        x1, x2 = _cloud(rng, b, n, int(rng.integers(0, 8))), _cloud(rng, b, m, int(rng.integers(0, 8)))
    if seed % 4 == 0:     # the second cloud a little off the first: queries in empty blocks
        x2 = x2 + np.float32(rng.choice([0.02, 0.3]))
    if seed % 7 == 0:     # one batch element of another family
        x1[0] = _cloud(rng, 1, n, 1)[0]
    x1, x2 = np.ascontiguousarray(x1, np.float32), np.ascontiguousarray(x2, np.float32)
    t1, t2 = torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda)
    lib = _lib.lib()
    search = lib.pp_debug_set_nmdistance_search
    search.argtypes = [ctypes.c_int]
    search.restype = None
    search(1)
    try:
        ref = [a.cpu().numpy() for a in nndistance(t1, t2)]
    finally:
        search(0)
    for launch in range(2):
        got = [a.cpu().numpy() for a in nndistance(t1, t2)]
        for g, e, what in zip(got, ref, ("dist1", "dist2", "idx1", "idx2")):
            assert np.array_equal(g, e), "seed %d launch %d (b=%d n=%d m=%d kind %d): %s differs at %d places" % (
                seed, launch, b, n, m, kind, what, int((g != e).sum()))
