"""GPU parity of the exact grid search (chamfer_grid.hip) against the CPU oracle on point
distributions chosen to break a spatial index: the outputs must be bit-identical to the brute
force's (indices AND distances), whatever path a query takes (grid hit, radius-2 shell, fallback
list, useless-grid fallback)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from pytorch_points_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _run(cuda, x1, x2, mode):
    """mode 2: the grid search wherever it is structurally possible (automatic mode leaves small problems
    to the brute force); 1: brute force"""
    from pytorch_points_amd import _lib
    from pytorch_points_amd.network.model_loss import nndistance
    setter = _lib.lib().pp_debug_set_nmdistance_search
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(mode)
    try:
        d1, d2, i1, i2 = nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda))
        torch.cuda.synchronize()
    finally:
        setter(0)
    return d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()


def _u(seed, shape):
    return S.uniform01(seed, shape).reshape(shape).astype(np.float32)


def _cases():
    n = 4096
    c = {}
    c["sphere"] = (S.unit_sphere(1, 2, n), S.unit_sphere(2, 2, n))
    c["sphere_ragged"] = (S.unit_sphere(3, 1, 2048), S.unit_sphere(4, 1, 5003))
    c["cube_volume"] = (_u(5, (2, n, 3)), _u(6, (2, n, 3)))
    blobs = (S.normal(7, (1, n, 3)) * 1e-3 + np.repeat(_u(8, (1, 8, 3)), n // 8, 1)).astype(np.float32)
    c["tight_blobs_vs_uniform"] = (blobs, _u(9, (1, n, 3)))
    out = _u(10, (1, n, 3)).copy()
    out[0, :50] += 40.0                                   # far outliers stretch the box
    c["outliers"] = (out, _u(11, (1, n, 3)))
    c["all_identical_refs"] = (_u(12, (1, n, 3)), np.full((1, n, 3), 0.25, np.float32))
    line = np.zeros((1, n, 3), np.float32)
    line[0, :, 0] = np.linspace(-1, 1, n, dtype=np.float32)
    c["collinear"] = (line, _u(13, (1, n, 3)))
    plane = _u(14, (1, n, 3)).copy()
    plane[..., 2] = 0.5
    c["planar"] = (plane, (plane[:, ::-1] + np.float32(1e-3)).copy())
    dup = S.unit_sphere(15, 1, n)
    dup[0, n // 2:] = dup[0, : n // 2]                    # every point twice: exact ties
    c["duplicates"] = (S.unit_sphere(16, 1, n), dup)
    c["self"] = (dup.copy(), dup.copy())                  # zero distances, ties
    c["huge_offset"] = (S.unit_sphere(17, 1, n) + np.float32(1e6), S.unit_sphere(18, 1, n) + np.float32(1e6))
    c["tiny_scale"] = (S.unit_sphere(19, 1, n) * np.float32(1e-18), S.unit_sphere(20, 1, n) * np.float32(1e-18))
    c["disjoint_far_apart"] = (_u(21, (1, n, 3)), _u(22, (1, n, 3)) + np.float32(100.0))
    lattice = np.stack(np.meshgrid(*[np.arange(16, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3)
    c["integer_lattice_ties"] = (lattice + np.float32(0.5), lattice.copy())   # 8-way exact ties
    mixed = np.concatenate([_u(23, (1, n // 2, 3)) * np.float32(1e-3), _u(24, (1, n // 2, 3)) * np.float32(50)], 1)
    c["two_scales"] = (mixed, _u(25, (1, n, 3)) * np.float32(50))
    # round 2: the paths for clouds that are not evenly sampled surfaces (crowded cells with their own grids, lane
    # cubes, the group search of far queries)
    c["gaussian"] = (S.normal(26, (2, n, 3)), S.normal(27, (2, n, 3)))
    cen = _u(28, (1, 8, 3)) * np.float32(2)
    same = (np.repeat(cen, n // 8, 1) + S.normal(29, (1, n, 3)) * np.float32(0.02)).astype(np.float32)
    same2 = (np.repeat(cen, n // 8, 1) + S.normal(30, (1, n, 3)) * np.float32(0.02)).astype(np.float32)
    c["blobs_same_places"] = (same, same2)                # crowded cells on both sides: second-level grids
    other = (np.repeat(_u(31, (1, 8, 3)) * np.float32(2), n // 8, 1) + S.normal(32, (1, n, 3)) * np.float32(0.02))
    c["blobs_other_places"] = (same, other.astype(np.float32))   # every query far from the references: groups
    half = _u(33, (1, n, 3)).copy()
    half[0, : n // 2] *= np.float32(1e-2)
    half2 = _u(34, (1, n, 3)).copy()
    half2[0, : n // 2] *= np.float32(1e-2)
    c["half_dense_half_sparse"] = (half, half2)
    c["sparse_volume_vs_surface"] = (_u(35, (1, n, 3)) * np.float32(2) - np.float32(1), S.unit_sphere(36, 1, n))
    c["dense_blob_vs_far_uniform"] = (half, _u(37, (1, n, 3)) + np.float32(3))
    return c


CASES = _cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_grid_search_equals_oracle(cuda, name):
    x1, x2 = [np.ascontiguousarray(a) for a in CASES[name]]
    exp = oracle.chamfer_forward(x1, x2)
    got = _run(cuda, x1, x2, 2)
    for g, e, what in zip(got, exp, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e), "%s: %s differs at %d places" % (name, what, int((g != e).sum()))
    brute = _run(cuda, x1, x2, 1)
    for g, e in zip(brute, exp):
        assert np.array_equal(g, e)


def test_grid_search_full_size_c2(cuda):
    """BASELINE config 2 through the default (grid) path == forced brute force, all 32 batch elements."""
    x1, x2 = S.unit_sphere(0, 32, 16384), S.unit_sphere(1, 32, 16384)
    a = _run(cuda, x1, x2, 2)
    b = _run(cuda, x1, x2, 1)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    e = oracle.chamfer_forward(x1[:2], x2[:2])
    for u, v in zip(a, e):
        assert np.array_equal(u[:2], v)


@pytest.mark.parametrize("kind", ["gaussian", "blobs8", "two_scales", "shapenet_like", "disjoint"])
def test_grid_search_full_size_other_distributions(cuda, kind):
    """The distributions of bench.py's other_distributions_fwd_ms at config-2 cloud size: grid == brute force,
    and two runs of the grid path agree (no order dependence in the in-kernel fallbacks)."""
    import bench
    x1, x2 = bench._distribution(kind, 0, 2, 16384), bench._distribution(kind, 1, 2, 16384)
    a = _run(cuda, x1, x2, 2)
    a2 = _run(cuda, x1, x2, 2)
    b = _run(cuda, x1, x2, 1)
    for u, v, w in zip(a, a2, b):
        assert np.array_equal(u, v) and np.array_equal(u, w)


@pytest.mark.parametrize("kind,n,m", [("two_scales", 40000, 40000), ("blobs8", 40000, 25000), ("gaussian", 100000, 70000)])
def test_grid_search_large_clustered_clouds(cuda, kind, n, m):
    """clouds beyond one build chunk (16384 points per workgroup pass) with crowded cells: the second-level grids
    built from the multi-chunk path, the group search over long rows; grid == brute force"""
    import bench
    x1, x2 = bench._distribution(kind, 0, 1, n), bench._distribution(kind, 1, 1, m)
    a = _run(cuda, x1, x2, 2)
    b = _run(cuda, x1, x2, 1)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    e = oracle.chamfer_forward(x1, x2)          # (the oracle, not only the every-pair kernel)
    for u, v, what in zip(a, e, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(u, v), "%s: %s differs at %d places" % (kind, what, int((u != v).sum()))


def test_grid_workspace_reused_across_shapes(cuda):
    for n, m in [(4096, 2048), (2048, 8192), (3000, 3000)]:
        x1, x2 = S.unit_sphere(n, 2, n), S.unit_sphere(m + 1, 2, m)
        got = _run(cuda, x1, x2, 2)
        exp = oracle.chamfer_forward(x1, x2)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)


def _run_labeled(cuda, x1, x2, l1, l2, mode):
    from pytorch_points_amd import _lib
    from pytorch_points_amd.network.model_loss import labeled_nndistance
    setter = _lib.lib().pp_debug_set_nmdistance_search
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(mode)
    try:
        d1, d2, i1, i2 = labeled_nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda),
                                            torch.from_numpy(l1).to(cuda), torch.from_numpy(l2).to(cuda))
        torch.cuda.synchronize()
    finally:
        setter(0)
    return d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()


def _labels(seed, shape, nlabels):
    return np.floor(S.uniform01(seed, shape).reshape(shape) * nlabels).astype(np.float32)


@pytest.mark.parametrize("labels", ["four", "one_missing", "all_same", "mostly_unique", "disjoint"])
@pytest.mark.parametrize("name", ["sphere", "sphere_ragged", "cube_volume", "tight_blobs_vs_uniform", "duplicates",
                                  "all_identical_refs", "huge_offset", "integer_lattice_ties", "outliers", "gaussian",
                                  "blobs_same_places", "blobs_other_places", "half_dense_half_sparse"])
def test_labeled_grid_search_equals_oracle(cuda, name, labels):
    """labeled Chamfer through the grid (the label filter inside the staged search, the labeled list
    fallback, idx -1 / dist 0 for queries without a partner) == oracle, bit for bit"""
    x1, x2 = [np.ascontiguousarray(a) for a in CASES[name]]
    s1, s2 = x1.shape[:2], x2.shape[:2]
    if labels == "four":
        l1, l2 = _labels(50, s1, 4), _labels(51, s2, 4)
    elif labels == "one_missing":
        l1, l2 = _labels(52, s1, 4), _labels(53, s2, 3)          # label 3 has no partner in cloud 2
    elif labels == "all_same":
        l1, l2 = np.full(s1, 7.0, np.float32), np.full(s2, 7.0, np.float32)
    elif labels == "mostly_unique":
        l1, l2 = _labels(54, s1, 3000), _labels(55, s2, 3000)    # most queries have no or one partner
    else:
        l1, l2 = _labels(56, s1, 2), _labels(57, s2, 2) + np.float32(10)   # nobody matches
    exp = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    got = _run_labeled(cuda, x1, x2, l1, l2, 2)
    for g, e, what in zip(got, exp, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e), "%s/%s: %s differs at %d places" % (name, labels, what, int((g != e).sum()))


@pytest.mark.parametrize("shape,shift", [((3, 4096, 4096), 1), ((3, 4096, 4096), 0), ((2, 4099, 5001), 0),
                                          ((2, 4099, 5001), 3), ((1, 2300, 2049), 2)])
def test_alignment_paths_forward_and_backward(cuda, shape, shift):
    """The grid build and the Chamfer backward each have a 16-byte-load kernel (aligned clouds, point
    counts that are multiples of 4) and a 4-byte one, picked by the host from pointer alignment and shape.
    Clouds carved out of a buffer at a 4-byte offset, and point counts that are not multiples of 4 (also not
    of 256: ragged last tile of the query kernel), must take the scalar paths and give the same bits."""
    from pytorch_points_amd import _lib
    from pytorch_points_amd.network.model_loss import nndistance
    b, n, m = shape
    x1, x2 = S.unit_sphere(41, b, n), S.unit_sphere(42, b, m)

    def carve(x):  # a contiguous (B, N, 3) view whose first element sits `shift` floats into an allocation
        buf = torch.empty(x.size + shift + 8, device=cuda)
        v = buf[shift:shift + x.size].view(x.shape)
        v.copy_(torch.from_numpy(x).to(cuda))
        assert v.is_contiguous() and v.data_ptr() % 16 == (4 * shift) % 16
        return v.requires_grad_(True)

    t1, t2 = carve(x1), carve(x2)
    setter = _lib.lib().pp_debug_set_nmdistance_search
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(2)
    try:
        d1, d2, i1, i2 = nndistance(t1, t2)
        g1 = torch.from_numpy(S.uniform01(43, (b, n)).reshape(b, n).astype(np.float32)).to(cuda)
        g2 = torch.from_numpy(S.uniform01(44, (b, m)).reshape(b, m).astype(np.float32)).to(cuda)
        torch.autograd.backward([d1, d2], [g1, g2])
        torch.cuda.synchronize()
    finally:
        setter(0)
    e = oracle.chamfer_forward(x1, x2)
    assert np.array_equal(i1.cpu().numpy(), e[1]) and np.array_equal(i2.cpu().numpy(), e[3])
    assert np.array_equal(d1.detach().cpu().numpy(), e[0]) and np.array_equal(d2.detach().cpu().numpy(), e[2])
    r1, r2 = oracle.chamfer_backward(x1, x2, g1.cpu().numpy(), g2.cpu().numpy(), e[1], e[3])
    # tolerance of the backward (same terms, summation order unspecified): 1e-5 relative
    assert np.allclose(t1.grad.cpu().numpy(), r1, rtol=1e-5, atol=1e-9)
    assert np.allclose(t2.grad.cpu().numpy(), r2, rtol=1e-5, atol=1e-9)


def test_alignment_paths_shared_grid_builds(cuda):
    """ball_query, three_nn and knn_points build their grids with the same two kernels: unaligned clouds
    and odd point counts against the oracle."""
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.ops import knn_points
    b, n, m = 2, 4099, 1027
    x, c = S.unit_sphere(51, b, n), S.unit_sphere(52, b, m)
    for shift in (0, 1):
        def carve(a):
            buf = torch.empty(a.size + shift + 8, device=cuda)
            v = buf[shift:shift + a.size].view(a.shape)
            v.copy_(torch.from_numpy(a).to(cuda))
            return v
        tx, tc = carve(x), carve(c)
        idx = sampling.ball_query(tc, tx, 0.2, 16)
        assert np.array_equal(idx.cpu().numpy(), oracle.ball_query(c, x, 0.2, 16))
        d2 = torch.empty(b, n, 3, device=cuda)
        i3 = torch.empty(b, n, 3, dtype=torch.int32, device=cuda)
        sampling.three_nn_wrapper(b, n, m, tx, tc, d2, i3)
        ed, ei = oracle.three_nn(x, c)
        assert np.array_equal(i3.cpu().numpy(), ei) and np.array_equal(d2.cpu().numpy(), ed)
        kn = knn_points(tc, tx, K=4)
        od, oi = oracle.knn(c, x, 4)
        assert np.array_equal(kn.idx.cpu().numpy().astype(np.int32), oi) and np.array_equal(kn.dists.cpu().numpy(), od)


def test_large_clouds_multi_chunk_build(cuda):
    """Clouds beyond 16384 points take the build's multi-chunk path (points re-read per pass, no cached cell
    slots) and put ~10x more points into every cell of the 32^3 grid: grid search == brute-force kernel ==
    oracle, bit for bit, on odd sizes."""
    x1, x2 = S.unit_sphere(61, 2, 40000), S.unit_sphere(62, 2, 70001)
    got = _run(cuda, x1, x2, 2)
    brute = _run(cuda, x1, x2, 1)
    for a, b in zip(got, brute):
        assert np.array_equal(a, b)
    e = oracle.chamfer_forward(x1, x2)
    assert np.array_equal(got[1], e[1]) and np.array_equal(got[3], e[3])
    assert np.array_equal(got[0], e[0]) and np.array_equal(got[2], e[2])


# ---- round 3: the unlabeled search as two kernels -- stage A by tiles (persistent workgroups, one LDS image of the
# reference layers a tile can touch), then the list kernel over what it left.  Every tile size, and the single-kernel
# form of round 2, against the oracle and against each other.
def _tile_knob(v):
    from pytorch_points_amd import _lib
    fn = _lib.lib().pp_debug_set_nmdistance_tile
    fn.argtypes = [ctypes.c_int]
    fn.restype = None
    fn(v)


@pytest.mark.parametrize("tile", [-1, 256, 512, 1024])
@pytest.mark.parametrize("shape", [(2, 4096, 4096), (1, 2048, 5003), (3, 4099, 2049), (1, 16384, 16384), (2, 8193, 8191)])
def test_stage_a_tile_sizes_equal_oracle(cuda, tile, shape):
    """ragged clouds (tiles and chunks that end inside a wave, clouds of different sizes, several batch elements),
    surfaces: what stage A serves itself"""
    b, n, m = shape
    x1, x2 = S.unit_sphere(100 + n, b, n), S.unit_sphere(200 + m, b, m)
    _tile_knob(tile)
    try:
        got = _run(cuda, x1, x2, 2)
    finally:
        _tile_knob(0)
    exp = oracle.chamfer_forward(x1, x2)
    assert np.array_equal(got[1], exp[1]) and np.array_equal(got[3], exp[3])
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[2], exp[2])


@pytest.mark.parametrize("tile", [256, 512, 1024])
@pytest.mark.parametrize("name", ["cube_volume", "gaussian", "blobs_other_places", "duplicates", "integer_lattice_ties",
                                  "planar", "outliers", "huge_offset", "all_identical_refs", "two_scales"])
def test_stage_a_leftovers_equal_brute_force(cuda, tile, name):
    """clouds stage A serves in part or not at all (images beyond its capacity, second-level grids, exact ties, far
    queries): whatever it leaves, the list kernel must settle with the brute force's bits"""
    x1, x2 = CASES[name]
    ref = _run(cuda, x1, x2, 1)
    _tile_knob(tile)
    try:
        got = _run(cuda, x1, x2, 2)
    finally:
        _tile_knob(0)
    for a, e in zip(got, ref):
        assert np.array_equal(a, e), name


def test_stage_a_serves_an_evenly_sampled_surface(cuda):
    """at the benchmark's configuration the list kernel must be left with about one query in a hundred (if this
    grows, the forward silently falls back to the slow path)"""
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import losses
    b, n = 4, 16384
    x1 = torch.from_numpy(S.unit_sphere(0, b, n)).to(cuda)
    x2 = torch.from_numpy(S.unit_sphere(1, b, n)).to(cuda)
    d1 = torch.empty(b, n, device=cuda); d2 = torch.empty(b, n, device=cuda)
    i1 = torch.empty(b, n, dtype=torch.int32, device=cuda); i2 = torch.empty(b, n, dtype=torch.int32, device=cuda)
    losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
    tot = (ctypes.c_uint * (2 * b))()
    fn = _lib.lib().pp_debug_nmdistance_pending
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    fn.restype = ctypes.c_int
    ws = [w for w in _lib.cached_workspaces("nmdistance") if w.device == x1.device][-1]
    assert fn(ws.data_ptr(), b, n, n, tot) == 0
    left = sum(tot) / (2.0 * b * n)
    assert left < 0.02, left


def _far_cases():
    """clouds far from each other (round 3: the group search cuts rows by the untrimmed box's own faces, member by
    member, and sifts candidates): boxes that end exactly at their extreme points, trimmed boxes whose outliers sit
    next to the far queries, clouds far from the origin against their extent, flat and degenerate extents"""
    n = 6000
    u = lambda seed, shape: _u(seed, shape)
    c = {}
    for k, off in enumerate([(5, 0, 0), (0, -7, 0), (3, 3, 3), (-2, 4, -9), (0.5, 0, 0), (1.02, 0, 0)]):
        c["offset_%s" % "_".join(str(v) for v in off)] = (u(500 + k, (2, n, 3)), u(520 + k, (2, n, 3)) + np.array(off, np.float32))
    # a trimmed box: 12 outliers of the reference cloud sit right beside the far query cloud
    ref = S.normal(540, (1, n, 3)).astype(np.float32) * np.float32(0.05)
    ref[0, :12] = u(541, (12, 3)) * np.float32(0.1) + np.float32(9.0)
    c["outliers_next_to_the_queries"] = (u(542, (1, n, 3)) * np.float32(0.2) + np.float32(9.0), ref)
    c["outliers_on_the_far_side"] = (u(543, (1, n, 3)) * np.float32(0.2) - np.float32(9.0), ref)
    # far from the origin against the extent (cell faces are rounded at the coordinates' magnitude)
    c["both_far_from_origin"] = (u(544, (1, n, 3)) + np.float32(2000.0), u(545, (1, n, 3)) + np.float32(2003.0))
    flat = u(546, (1, n, 3)).copy()
    flat[..., 2] = 0.25
    c["plane_vs_far_plane"] = (flat, flat[:, ::-1].copy() + np.array([0, 0, 4], np.float32))
    line = np.zeros((1, n, 3), np.float32)
    line[..., 0] = u(547, (1, n))
    c["line_vs_far_cloud"] = (line, u(548, (1, n, 3)) + np.array([2, 2, 2], np.float32))
    c["small_far_big"] = (u(549, (1, 700, 3)) * np.float32(0.01), u(550, (1, n, 3)) * np.float32(30.0) + np.float32(40.0))
    return c


FAR = _far_cases()


@pytest.mark.parametrize("name", sorted(FAR))
def test_far_clouds_equal_oracle(cuda, name):
    x1, x2 = [np.ascontiguousarray(a) for a in FAR[name]]
    exp = oracle.chamfer_forward(x1, x2)
    got = _run(cuda, x1, x2, 2)
    for g, e, what in zip(got, exp, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e), "%s: %s differs at %d places" % (name, what, int((g != e).sum()))


# ---- clouds of config 2's size class (8192 .. 17408 points: the sizes the build's one-chunk path sorts: surfaces, volumes,
# ties, a cloud sorted along the build slabs' axis, outliers, a NaN, mixed batches): the default search's four outputs
# are the every-pair kernel's, bit for bit, launch after launch on the same workspace.
def _class2_cases():
    u = lambda seed, shape: S.uniform01(seed, shape).reshape(shape).astype(np.float32)
    n = 16384
    c = {}
    c["surface"] = (S.unit_sphere(700, 2, n), S.unit_sphere(701, 2, n), "served")
    c["surface_ragged"] = (S.unit_sphere(702, 3, 12292), S.unit_sphere(703, 3, 16380), "served")
    c["surface_smallest"] = (S.unit_sphere(704, 1, 8192), S.unit_sphere(705, 1, 12288), None)
    same = S.unit_sphere(706, 1, n)
    c["same_cloud"] = (same, same.copy(), "served")                     # every distance zero
    dup = S.unit_sphere(707, 1, n)
    dup[0, n // 2:] = dup[0, : n // 2]                                    # every point twice: exact ties across groups
    c["duplicates"] = (S.unit_sphere(708, 1, n), dup, None)
    lat = np.round(S.unit_sphere(709, 1, n) * 24).astype(np.float32)     # a lattice: ties everywhere
    c["lattice"] = (lat, np.round(S.unit_sphere(710, 1, n) * 24).astype(np.float32), None)
    zs = S.unit_sphere(711, 1, n)
    zs = zs[:, np.argsort(zs[0, :, 2])]                                   # sorted along the slabs' axis
    c["sorted_in_z"] = (np.ascontiguousarray(zs), S.unit_sphere(712, 1, n), None)
    c["volume"] = (u(713, (2, n, 3)), u(714, (2, n, 3)), "declined")
    c["gaussian"] = (S.normal(715, (1, n, 3)).astype(np.float32), S.normal(716, (1, n, 3)).astype(np.float32), None)
    flat = u(717, (1, n, 3)).copy()
    flat[..., 2] = 0.3
    c["plane"] = (flat, flat[:, ::-1].copy() + np.float32(1e-3), "declined")
    c["far_apart"] = (S.unit_sphere(718, 1, n), S.unit_sphere(719, 1, n) + np.float32(5.0), None)
    out = S.unit_sphere(720, 1, n)
    out[0, 5::1000] *= np.float32(30.0)                                   # outliers the sampled box does not see
    c["outliers"] = (out, S.unit_sphere(721, 1, n), None)
    mixed1 = np.concatenate([S.unit_sphere(722, 1, n), u(723, (1, n, 3)), S.unit_sphere(724, 1, n)])
    mixed2 = np.concatenate([S.unit_sphere(725, 1, n), u(726, (1, n, 3)), S.unit_sphere(727, 1, n)])
    c["mixed_batch"] = (mixed1, mixed2, "mixed")
    nan = S.unit_sphere(728, 2, n)
    nan[1, 777, 1] = np.nan
    c["a_nan"] = (nan, S.unit_sphere(729, 2, n), "mixed_nan")
    c["not_a_multiple_of_four"] = (S.unit_sphere(730, 1, 16383), S.unit_sphere(731, 1, n), "absent")
    # surfaces that are NOT spread evenly along z (round 5: the build then deals the slabs' layers by a histogram, and
    # stage A's tiles read the layer table those slabs write): three quarters of a sphere's points in its lower cap; a
    # cone (its layers grow with z)
    def uneven(seed):
        p = S.unit_sphere(seed, 2, n).copy()
        low = p[:, : 3 * n // 4]
        low[..., 2] = -np.abs(low[..., 2]) * np.float32(0.5) - np.float32(0.5)
        low[..., :2] *= np.sqrt(np.maximum(1.0 - low[..., 2:3] ** 2, 0.0)).astype(np.float32) / np.maximum(
            np.linalg.norm(low[..., :2], axis=-1, keepdims=True), 1e-6).astype(np.float32)
        return np.ascontiguousarray(p.astype(np.float32))
    c["surface_uneven_in_z"] = (uneven(732), uneven(733), None)
    def cone(seed):
        u = S.uniform01(seed, (2, n, 2)).reshape(2, n, 2).astype(np.float32)
        zz = np.sqrt(u[..., 0])                                            # area-uniform on the cone
        th = u[..., 1] * np.float32(6.2831853)
        return np.ascontiguousarray(np.stack([zz * np.cos(th), zz * np.sin(th), zz], -1).astype(np.float32))
    c["cone"] = (cone(734), cone(735), None)
    return c


CLASS2 = _class2_cases()


@pytest.mark.parametrize("name", sorted(CLASS2))
def test_config2_class_clouds_equal_brute_force(cuda, name):
    x1, x2, _ = CLASS2[name]
    x1, x2 = np.ascontiguousarray(x1), np.ascontiguousarray(x2)
    ref = _run(cuda, x1, x2, 1)
    for rep in range(2):                                   # (the workspace is reused launch after launch)
        got = _run(cuda, x1, x2, 2)
        for g, e, what in zip(got, ref, ["dist1", "idx1", "dist2", "idx2"]):
            same = np.array_equal(g, e, equal_nan=True)
            assert same, "%s (launch %d): %s differs at %d places" % (name, rep, what, int((g != e).sum()))


# ---- round 5: the balls behind stage A (lane_ball_search, wave_pooled_ball_search): clouds of mixed dimension -- thin
# faces with a sparse interior, where a query's block holds a candidate too far to settle it -- at several sizes, far
# from the origin (the ball is measured in cells from the query's own cell coordinates), with outliers clamped into the
# rim cells the balls reach, with ties; unlabeled (the pooled form) and labeled (the lane form): the every-pair kernel's
# outputs, bit for bit.
def _object_like(seed, b, n, offset=0.0, scale=1.0):
    u = _u(seed, (b, n, 3)) - np.float32(0.5)
    q = n // 4
    u[:, :q, 2] = -0.5                                              # a face in the lowest layer
    u[:, q:2 * q, 0] = 0.2                                          # a wall
    th = _u(seed + 1, (b, q)) * np.float32(6.283)
    u[:, 2 * q:3 * q, 0] = np.float32(0.3) * np.cos(th)
    u[:, 2 * q:3 * q, 1] = np.float32(0.3) * np.sin(th)             # a cylinder; the last quarter fills the volume
    return (u * np.float32(scale) + np.float32(offset)).astype(np.float32)


def _ball_cases():
    c = {}
    c["object_16384"] = (_object_like(800, 2, 16384), _object_like(802, 2, 16384))
    c["object_ragged"] = (_object_like(804, 2, 9001), _object_like(806, 2, 12345))
    c["object_small"] = (_object_like(808, 3, 3000), _object_like(810, 3, 4096))
    c["object_far_from_origin"] = (_object_like(812, 1, 16384, offset=1000.0), _object_like(814, 1, 16384, offset=1000.0))
    c["object_tiny_scale"] = (_object_like(816, 1, 16384, scale=1e-3), _object_like(818, 1, 16384, scale=1e-3))
    out = _object_like(820, 1, 16384)
    out[0, 3::500] *= np.float32(40.0)                               # outliers: clamped into the rim cells
    c["object_with_outliers"] = (_object_like(822, 1, 16384), out)
    c["queries_beside_the_box"] = (_object_like(824, 1, 8192) + np.float32([0.6, 0.0, 0.0]), _object_like(826, 1, 16384))
    lat = np.round(_object_like(828, 1, 16384) * 20).astype(np.float32)
    c["object_on_a_lattice"] = (lat, np.round(_object_like(830, 1, 16384) * 20).astype(np.float32))
    vol = _u(832, (2, 16384, 3))
    c["volume_vs_object"] = (vol - np.float32(0.5), _object_like(834, 2, 16384))
    sparse = _u(836, (1, 2048, 3)) - np.float32(0.5)
    c["sparse_queries_dense_object"] = (sparse, _object_like(838, 1, 16384))
    return c


BALL = _ball_cases()


@pytest.mark.parametrize("name", sorted(BALL))
def test_ball_stage_clouds_equal_brute_force(cuda, name):
    x1, x2 = (np.ascontiguousarray(a) for a in BALL[name])
    ref = _run(cuda, x1, x2, 1)
    got = _run(cuda, x1, x2, 2)
    for g, e, what in zip(got, ref, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e, equal_nan=True), "%s: %s differs at %d places" % (name, what, int((g != e).sum()))


@pytest.mark.parametrize("name", ["object_ragged", "object_with_outliers", "object_on_a_lattice", "volume_vs_object"])
def test_ball_stage_clouds_labeled_equal_oracle(cuda, name):
    x1, x2 = (np.ascontiguousarray(a[:1, :6000]) for a in BALL[name])
    l1, l2 = _labels(60, x1.shape[:2], 3), _labels(61, x2.shape[:2], 3)
    exp = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    got = _run_labeled(cuda, x1, x2, l1, l2, 2)
    for g, e, what in zip(got, exp, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e), "%s: %s differs at %d places" % (name, what, int((g != e).sum()))


@pytest.mark.parametrize("nlab", [16, 64])
@pytest.mark.parametrize("name", ["object_ragged", "volume_vs_object"])
def test_ball_stage_many_labels_equal_oracle(cuda, name, nlab):
    """many labels: a block seldom holds the query's label, nearly every lane of a wave is open and the pooled stages
    flush their piece lists several times per wave"""
    x1, x2 = (np.ascontiguousarray(a[:1, :5000]) for a in BALL[name])
    l1, l2 = _labels(62, x1.shape[:2], nlab), _labels(63, x2.shape[:2], nlab)
    exp = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    got = _run_labeled(cuda, x1, x2, l1, l2, 2)
    for g, e, what in zip(got, exp, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e), "%s/%d: %s differs at %d places" % (name, nlab, what, int((g != e).sum()))


@pytest.mark.parametrize("n,m", [(16384, 2048), (2048, 16384), (12000, 3000)])
def test_ball_stage_sparse_reference_equal_brute_force(cuda, n, m):
    """a dense query cloud against a sparse reference (and the reverse): every query's block is nearly empty, the balls
    and the cubes of radius 1 and 2 carry the search"""
    x1 = _u(840, (2, n, 3)) - np.float32(0.5)
    x2 = _object_like(842, 2, m)
    ref = _run(cuda, x1, x2, 1)
    got = _run(cuda, x1, x2, 2)
    for g, e, what in zip(got, ref, ["dist1", "idx1", "dist2", "idx2"]):
        assert np.array_equal(g, e, equal_nan=True), "%d/%d: %s differs at %d places" % (n, m, what, int((g != e).sum()))


# ---- round 6: directions no search can prune are routed to the every-pair kernel (the stage-A launch decides on the
# device; the every-pair launch follows once the host has seen a routed direction).  Whatever is routed, and whichever
# kernel serves it, the outputs are the every-pair kernel's, bit for bit.
def _adversarial(name, b, n):
    rng = np.random.default_rng(950)
    if name == "shell_vs_core":
        a = rng.standard_normal((b, n, 3)); a /= np.linalg.norm(a, axis=-1, keepdims=True)
        return a.astype(np.float32), (rng.standard_normal((b, n, 3)) * 1e-3).astype(np.float32)
    if name == "identical":
        return np.full((b, n, 3), 0.25, np.float32), np.full((b, n, 3), 0.75, np.float32)
    if name == "shells":
        a = rng.standard_normal((b, n, 3)); a /= np.linalg.norm(a, axis=-1, keepdims=True)
        c = rng.standard_normal((b, n, 3)); c /= np.linalg.norm(c, axis=-1, keepdims=True)
        return a.astype(np.float32), (c * 0.5).astype(np.float32)
    if name == "mixed":          # element 0 ordinary, element 1 a shell against its core: routed per direction
        a = S.unit_sphere(951, b, n); c = S.unit_sphere(952, b, n)
        c[1] = (rng.standard_normal((n, 3)) * 1e-3).astype(np.float32)
        return a, c
    return S.unit_sphere(953, b, n), S.unit_sphere(954, b, n)


@pytest.mark.parametrize("name", ["shell_vs_core", "identical", "shells", "mixed", "sphere"])
def test_routed_directions_equal_brute_force(cuda, name):
    from pytorch_points_amd import _lib
    route = _lib.lib().pp_debug_set_nmdistance_routing
    route.argtypes = [ctypes.c_int]
    route.restype = None
    x1, x2 = _adversarial(name, 3, 8192)
    ref = _run(cuda, x1, x2, 1)
    try:
        for mode, reps in ((1, 1), (2, 2), (0, 3)):      # never routed / the every-pair launch always / automatic (the
            route(mode)                                   # launch follows from the second call on where something is routed)
            for rep in range(reps):
                got = _run(cuda, x1, x2, 2)
                for g, e, what in zip(got, ref, ["dist1", "idx1", "dist2", "idx2"]):
                    assert np.array_equal(g, e, equal_nan=True), "%s routing mode %d call %d: %s differs at %d places" % (
                        name, mode, rep, what, int((g != e).sum()))
    finally:
        route(0)


# ---- round 6: the group search takes the non-empty cell rows only (a per-set row bitmap written in the stage-A launch's
# tail).  Clouds whose grids are mostly empty rows -- clusters at different places, a line, a few outliers that stretch
# the box -- searched with the bitmap (default) and without it: both are the every-pair kernel's bits.
def _sparse_row_cases():
    c = {}
    rng = np.random.default_rng(77)

    def blobs(seed, b, n, k, sigma, span):
        r = np.random.default_rng(seed)
        cen = r.random((b, k, 3), dtype=np.float32) * np.float32(span)
        pick = r.integers(0, k, n)
        return (cen[:, pick] + r.standard_normal((b, n, 3)).astype(np.float32) * np.float32(sigma)).astype(np.float32)

    c["blobs8_other_places"] = (blobs(1, 3, 16384, 8, 0.02, 2.0), blobs(2, 3, 16384, 8, 0.02, 2.0))
    c["blobs3_vs_blobs40"] = (blobs(3, 2, 9000, 3, 0.01, 1.0), blobs(4, 2, 12000, 40, 0.005, 1.0))
    c["one_blob_vs_far_blob"] = (blobs(5, 2, 8192, 1, 0.03, 1.0), blobs(6, 2, 8192, 1, 0.03, 1.0) + np.float32(3.0))
    line = np.zeros((2, 8192, 3), np.float32)
    line[..., 0] = rng.random((2, 8192), dtype=np.float32)
    line[..., 1] = line[..., 0] * np.float32(0.5)
    c["diagonal_line_vs_blobs"] = (line, blobs(7, 2, 10000, 5, 0.02, 1.5))
    stretched = blobs(8, 2, 8192, 2, 0.01, 0.3)
    stretched[:, :6] = rng.random((2, 6, 3), dtype=np.float32) * np.float32(4.0)
    c["two_blobs_and_six_outliers"] = (blobs(9, 2, 8192, 4, 0.02, 2.0) + np.float32(1.0), stretched)
    c["ragged_blobs"] = (blobs(10, 1, 16383, 6, 0.02, 2.0), blobs(11, 1, 8195, 6, 0.02, 2.0))
    return c


SPARSE_ROWS = _sparse_row_cases()


@pytest.mark.parametrize("name", sorted(SPARSE_ROWS))
def test_group_search_row_bitmap_equals_brute_force(cuda, name):
    from pytorch_points_amd import _lib
    knob = _lib.lib().pp_debug_set_nmdistance_row_bitmap
    knob.argtypes = [ctypes.c_int]
    knob.restype = None
    x1, x2 = [np.ascontiguousarray(a) for a in SPARSE_ROWS[name]]
    ref = _run(cuda, x1, x2, 1)
    try:
        for off in (0, 1, 0):
            knob(off)
            got = _run(cuda, x1, x2, 2)
            for g, e, what in zip(got, ref, ["dist1", "idx1", "dist2", "idx2"]):
                assert np.array_equal(g, e), "%s (row bitmap %s): %s differs at %d places" % (
                    name, "off" if off else "on", what, int((g != e).sum()))
    finally:
        knob(0)
