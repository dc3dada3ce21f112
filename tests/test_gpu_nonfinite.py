"""Non-finite coordinates (NaN, +-inf) in queries and references: the grid kernels must terminate and
agree with the scan / brute-force kernels bit for bit (the comparison is GPU path against GPU path: the
reference leaves the result for such inputs to the order of its comparisons, which both paths share)."""
import ctypes

import numpy as np
import pytest
import torch

from pytorch_points_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _knob(name):
    from pytorch_points_amd import _lib
    f = getattr(_lib.lib(), name)
    f.argtypes = [ctypes.c_int]
    f.restype = None
    return f


def _poison(x, seed, where):
    """scatter NaN / inf / -inf into a copy of x (b, n, 3)"""
    rng = np.random.default_rng(seed)
    x = x.copy()
    b, n, _ = x.shape
    for val in where:
        k = rng.integers(0, n, 5)
        x[rng.integers(0, b), k, rng.integers(0, 3, 5)] = val
    return x


CASES = [("nan_queries", [np.nan], []), ("inf_queries", [np.inf, -np.inf], []), ("nan_refs", [], [np.nan]),
         ("inf_refs", [], [np.inf, -np.inf]), ("everything", [np.nan, np.inf], [np.nan, -np.inf])]


def _eq(a, b):
    return torch.equal(torch.nan_to_num(a.float(), nan=-7.0, posinf=-8.0, neginf=-9.0),
                       torch.nan_to_num(b.float(), nan=-7.0, posinf=-8.0, neginf=-9.0))


@pytest.mark.parametrize("name,qbad,rbad", CASES)
def test_nonfinite_chamfer(cuda, name, qbad, rbad):
    from pytorch_points_amd.network.model_loss import nndistance
    x1 = _poison(S.unit_sphere(300, 2, 4096), 1, qbad)
    x2 = _poison(S.unit_sphere(301, 2, 5000), 2, rbad)
    mode = _knob("pp_debug_set_nmdistance_search")
    out = []
    try:
        for m in (2, 1):
            mode(m)
            out.append(nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda)))
            torch.cuda.synchronize()
    finally:
        mode(0)
    for a, b, what in zip(out[0], out[1], ("dist1", "dist2", "idx1", "idx2")):
        assert _eq(a, b), "%s: %s differs between grid and brute force" % (name, what)


@pytest.mark.parametrize("name,qbad,rbad", CASES)
def test_nonfinite_ball_three_nn_knn(cuda, name, qbad, rbad):
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.ops import knn_points
    q = _poison(S.unit_sphere(302, 2, 1500), 3, qbad)
    r = _poison(S.unit_sphere(303, 2, 4096), 4, rbad)
    tq, tr = torch.from_numpy(q).to(cuda), torch.from_numpy(r).to(cuda)
    for knob, fn in (("pp_debug_set_ball_query_search", lambda: (sampling.ball_query(tq, tr, 0.15, 24),)),
                     ("pp_debug_set_knn_search", lambda: knn_points(tq, tr, K=6)[:2])):
        k = _knob(knob)
        out = []
        try:
            for m in (2 if "ball" in knob else 0, 1):
                k(m)
                out.append(fn())
                torch.cuda.synchronize()
        finally:
            k(0)
        for a, b in zip(out[0], out[1]):
            assert _eq(a, b), "%s: %s differs between grid and scan" % (name, knob)
    k = _knob("pp_debug_set_three_nn_search")
    out = []
    try:
        for m in (0, 1):
            k(m)
            d2 = torch.empty(2, 1500, 3, device=cuda)
            idx = torch.empty(2, 1500, 3, dtype=torch.int32, device=cuda)
            sampling.three_nn_wrapper(2, 1500, 4096, tq, tr, d2, idx)
            torch.cuda.synchronize()
            out.append((d2, idx))
    finally:
        k(0)
    assert _eq(out[0][0], out[1][0]) and _eq(out[0][1], out[1][1]), "%s: three_nn differs" % name


# Round 6: non-finite points no longer cost a set its grid (the plain builds recompute box and moments over the finite
# points, out of line): larger sets -- the LDS-sorted build (16384 aligned points), the general build in one chunk
# (unaligned) and in several chunks (40001 points) -- with a handful of NaN / inf points among the references AND the
# queries; the grid paths against the scan kernels, bit for bit, and the grid really used (no fallback to the scan).
@pytest.mark.parametrize("nref,nq", [(16384, 4096), (16383, 3001), (40001, 5000)])
def test_nonfinite_points_keep_the_grid_larger_sets(cuda, nref, nq):
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.ops import knn_points
    q = _poison(S.unit_sphere(310, 2, nq), 5, [np.nan, np.inf])
    r = _poison(S.unit_sphere(311, 2, nref), 6, [np.nan, -np.inf, np.inf])
    tq, tr = torch.from_numpy(q).to(cuda), torch.from_numpy(r).to(cuda)
    for knob, fn in (("pp_debug_set_ball_query_search", lambda: (sampling.ball_query(tq, tr, 0.08, 16),)),
                     ("pp_debug_set_knn_search", lambda: knn_points(tq, tr, K=5)[:2])):
        k = _knob(knob)
        out = []
        try:
            for m in (2 if "ball" in knob else 0, 1):
                k(m)
                out.append(fn())
                torch.cuda.synchronize()
        finally:
            k(0)
        for a, b in zip(out[0], out[1]):
            assert _eq(a, b), "%d refs: %s differs between grid and scan" % (nref, knob)
    k = _knob("pp_debug_set_three_nn_search")
    out = []
    try:
        for m in (0, 1):
            k(m)
            d2 = torch.empty(2, nq, 3, device=cuda)
            idx = torch.empty(2, nq, 3, dtype=torch.int32, device=cuda)
            sampling.three_nn_wrapper(2, nq, nref, tq, tr, d2, idx)
            torch.cuda.synchronize()
            out.append((d2, idx))
    finally:
        k(0)
    assert _eq(out[0][0], out[1][0]) and _eq(out[0][1], out[1][1]), "%d refs: three_nn differs" % nref
