"""GPU parity: FPS, gather, ball_query, group_points, three_nn, three_interpolate vs the CPU oracle
and vs torch identities.  Indices bit-exact; gathers bit-exact; scatter-adds (fp32
atomics) within 1e-5."""
import numpy as np
import pytest
import torch

import oracle
from pytorch_points_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


# ---------------------------------------------------------------- furthest point sampling + gather
_FPS_FORMS = {"default": 0, "single_block": 1, "cluster": 2, "bucket": 3}


def _fps_form(name):
    import ctypes
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_fps_v1
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(_FPS_FORMS[name])


def _fps_chain(form):
    import ctypes
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_fps_bucket_chain
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(form)


@pytest.fixture(params=["batched", "one_pick"])
def bucket_chain(request):
    """the bucketed kernel's two chains: several independent picks per barrier round (default) and one pick per
    round -- the same picks and temp from both"""
    _fps_chain({"batched": 2, "one_pick": 1}[request.param])
    yield request.param
    _fps_chain(0)


@pytest.fixture(params=["auto", "in_kernel"])
def bucket_sort(request):
    """where the bucketed kernel's counting sort runs: inside the kernel (its one workgroup), or as five short launches
    over the whole chip in front of it (the library's choice from 32768 points) -- any order inside a cell is correct"""
    import ctypes
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_fps_bucket_sort
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter({"auto": 0, "in_kernel": 1}[request.param])
    yield request.param
    setter(0)


@pytest.fixture(params=["default", "cluster", "single_block"])
def fps_path(request, cuda):
    """Run every FPS test on the three decompositions: the operator's own choice (the bucketed kernel where it
    applies: N >= 2048 and 32 or more picks), the CU-cluster kernel and the one-workgroup-per-batch kernel over all
    points; afterwards no inter-workgroup wait may have timed out."""
    from pytorch_points_amd._ext import sampling
    _fps_form(request.param)
    yield request.param
    _fps_form("default")
    assert sampling.furthest_sampling_status(cuda) == 0


@pytest.mark.parametrize("b,n,m,seed", [(2, 2048, 256, 0), (1, 300, 64, 7), (1, 5000, 128, 0),
                                        (3, 1024, 1024, 3), (1, 70, 70, 0), (2, 1, 1, 0), (1, 513, 40, 512)])
def test_fps_matches_oracle(cuda, fps_path, b, n, m, seed):
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    x = S.unit_sphere(10, b, n)
    idx, pc = furthest_point_sample(_t(x, cuda), m, NCHW=False, seedIdx=seed)
    e_idx, _ = oracle.furthest_sampling(x, m, seed)
    assert idx.dtype == torch.int32 and idx.shape == (b, m) and pc.shape == (b, m, 3)
    assert np.array_equal(idx.cpu().numpy(), e_idx)
    assert np.array_equal(pc.cpu().numpy(), np.take_along_axis(x, e_idx[..., None].astype(np.int64), 1))
    assert (idx[:, 0] == seed).all()
    # NCHW layout: (B,3,N) in, (B,3,npoint) out
    idx2, pc2 = furthest_point_sample(_t(x, cuda).transpose(1, 2).contiguous(), m, NCHW=True, seedIdx=seed)
    assert torch.equal(idx2, idx) and torch.equal(pc2, pc.transpose(1, 2))


def test_fps_ties_and_degenerate(cuda, fps_path):
    """Exact ties (duplicated points, all-equal points) follow the reference's thread-order rule."""
    from pytorch_points_amd._ext import sampling
    x = S.unit_sphere(11, 2, 1500)
    x[:, 700:1400] = x[:, :700]          # every point of the first 700 has an exact twin
    x[1, :] = x[1, 0]                    # batch 1: all points identical -> index 0 repeated
    m = 200
    xt = _t(x, cuda)
    idx = torch.empty(2, m, dtype=torch.int32, device=cuda)
    temp = torch.full((2, 1500), 1e10, dtype=torch.float32, device=cuda)
    out = sampling.furthest_sampling(m, 5, xt, temp, idx)
    assert out is idx
    e_idx, e_temp = oracle.furthest_sampling(x, m, 5)
    assert np.array_equal(idx.cpu().numpy(), e_idx)
    assert np.array_equal(temp.cpu().numpy(), e_temp)     # temp is an in/out argument
    assert (e_idx[1, 1:] == 0).all()


def _fps_clouds(n):
    """clouds the bucketed kernel's pruning is weakest or its sort most lopsided on"""
    rng = np.random.default_rng(77)
    out = {}
    out["sphere"] = S.unit_sphere(20, 1, n)[0]
    g = S.normal(21, (n, 3))
    out["gaussian"] = g
    c = rng.normal(size=(8, 3)).astype(np.float32) * 3
    out["blobs"] = (c[rng.integers(0, 8, n)] + 0.01 * S.normal(22, (n, 3))).astype(np.float32)
    two = S.unit_sphere(23, 1, n)[0].copy()
    two[: n // 2] = two[: n // 2] * 0.01 + 0.3            # half the points in 1e-6 of the volume
    out["two_scales"] = two
    pl = S.normal(24, (n, 3)); pl[:, 2] = 0.25
    out["plane"] = pl
    ln = np.zeros((n, 3), np.float32); ln[:, 0] = S.normal(25, (n,)); ln[:, 1] = 2 * ln[:, 0]
    out["line"] = ln
    lat = np.stack(np.meshgrid(*[np.arange(16, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    out["lattice"] = np.ascontiguousarray(lat[rng.integers(0, len(lat), n)])       # exact ties everywhere, duplicates
    out["identical"] = np.full((n, 3), 0.5, np.float32)
    far = S.unit_sphere(26, 1, n)[0] + np.float32(1000.0)
    out["offset_1000"] = far.astype(np.float32)
    outl = S.unit_sphere(27, 1, n)[0].copy(); outl[::997] *= 500.0
    out["outliers"] = outl
    return out


@pytest.mark.parametrize("n,m,seed", [(2048, 64, 0), (5000, 300, 7), (4099, 4099, 4098), (16384, 700, 3), (40000, 1500, 1)])
def test_fps_bucketed_kernel_on_hard_clouds(cuda, bucket_chain, bucket_sort, n, m, seed):
    """The bucketed kernel (every step only visits the buckets the pick can change) against the oracle: picks AND
    temp, on clouds with exact ties, duplicates, degenerate extents, clusters and outliers."""
    from pytorch_points_amd._ext import sampling
    clouds = _fps_clouds(n)
    names = sorted(clouds)
    x = np.stack([clouds[k] for k in names]).astype(np.float32)
    e_idx, e_temp = oracle.furthest_sampling(x, m, seed)
    _fps_form("bucket")
    try:
        idx = torch.empty(len(names), m, dtype=torch.int32, device=cuda)
        temp = torch.full((len(names), n), 1e10, dtype=torch.float32, device=cuda)
        pts = torch.empty(len(names), m, 3, device=cuda)
        sampling.furthest_sampling(m, seed, _t(x, cuda), temp, idx, pts, False)
    finally:
        _fps_form("default")
    got = idx.cpu().numpy()
    for i, k in enumerate(names):
        assert np.array_equal(got[i], e_idx[i]), (k, int(np.argmax(got[i] != e_idx[i])))
        assert np.array_equal(temp[i].cpu().numpy(), e_temp[i]), k
    assert np.array_equal(pts.cpu().numpy(), np.take_along_axis(x, e_idx[..., None].astype(np.int64), 1))


def test_fps_bucketed_kernel_honours_the_incoming_temp(cuda, bucket_chain):
    """temp is an in/out argument (ref sampling_cuda.cu:190,204-205): whatever the caller passes bounds every
    point's distance from the start -- also for the bucket maxima the pruning relies on"""
    from pytorch_points_amd._ext import sampling
    b, n, m = 3, 6000, 200
    x = S.unit_sphere(30, b, n)
    t0 = (np.abs(S.normal(31, (b, n))) * 0.05).astype(np.float32)
    t0[1] = 1e10
    t0[2, ::3] = 0.0
    e_idx, e_temp = oracle.furthest_sampling(x, m, 11, temp=t0)
    _fps_form("bucket")
    try:
        idx = torch.empty(b, m, dtype=torch.int32, device=cuda)
        temp = _t(t0, cuda)
        sampling.furthest_sampling(m, 11, _t(x, cuda), temp, idx)
    finally:
        _fps_form("default")
    assert np.array_equal(idx.cpu().numpy(), e_idx)
    assert np.array_equal(temp.cpu().numpy(), e_temp)


def test_fps_bucketed_kernel_beyond_65536_points(cuda, bucket_sort):
    """N > 65536: buckets of 128 or more points (at most 1024 buckets, one per thread)"""
    from pytorch_points_amd._ext import sampling
    for n, m in [(70000, 40), (200001, 33)]:
        x = S.unit_sphere(32, 1, n)
        e_idx, e_temp = oracle.furthest_sampling(x, m, 5)
        _fps_form("bucket")
        try:
            idx = torch.empty(1, m, dtype=torch.int32, device=cuda)
            temp = torch.full((1, n), 1e10, dtype=torch.float32, device=cuda)
            sampling.furthest_sampling(m, 5, _t(x, cuda), temp, idx)
        finally:
            _fps_form("default")
        assert np.array_equal(idx.cpu().numpy(), e_idx)
        assert np.array_equal(temp.cpu().numpy(), e_temp)


def test_fps_large_n_paths(cuda, fps_path):
    """N > 1024 (several points per thread) and N > 65536 (temp kept in global memory)."""
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    for n, m in [(20000, 64), (70000, 24)]:
        x = S.unit_sphere(12, 1, n)
        idx, _ = furthest_point_sample(_t(x, cuda), m, NCHW=False)
        e_idx, _ = oracle.furthest_sampling(x, m, 0)
        assert np.array_equal(idx.cpu().numpy(), e_idx)


def test_fps_config3_full_size(cuda):
    """BASELINE config 3 (B=16, N=65536 -> 4096): the operator's choice (the bucketed kernel) == the cluster kernel on
    every batch element, == the single-block kernel on three, == oracle on one; repeated calls reuse the workspace
    safely."""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    B, N, m = 16, 65536, 4096
    x = S.unit_sphere(13, B, N)
    xt = _t(x, cuda)
    runs = [furthest_point_sample(xt, m, NCHW=False)[0] for _ in range(3)]
    assert sampling.furthest_sampling_status(cuda) == 0
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    _fps_form("single_block")
    try:
        v1, _ = furthest_point_sample(xt[:3].contiguous(), m, NCHW=False)
        _fps_form("cluster")
        v2, _ = furthest_point_sample(xt, m, NCHW=False)
    finally:
        _fps_form("default")
    assert torch.equal(runs[0][:3], v1) and torch.equal(runs[0], v2)
    assert sampling.furthest_sampling_status(cuda) == 0
    e_idx, _ = oracle.furthest_sampling(x[:1], m, 0)
    assert np.array_equal(runs[0][:1].cpu().numpy(), e_idx)
    for b in range(B):
        assert len(set(runs[0][b].tolist())) == m


def test_fps_cluster_beside_other_work(cuda):
    """The cluster kernel needs its workgroups co-resident; with another stream keeping the chip
    busy they may be admitted late -- the bounded waits must ride that out (no timeout, same picks)."""
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    B, N, m = 8, 32768, 512
    x = S.unit_sphere(14, B, N)
    xt = _t(x, cuda)
    _fps_form("cluster")
    ref, _ = furthest_point_sample(xt, m, NCHW=False)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=cuda)
    outs = []
    for rep in range(3):
        with torch.cuda.stream(side):
            for _ in range(20):
                a = torch.tanh(a @ a * 1e-3)          # a few ms of chip-filling work per launch
        idx, _ = furthest_point_sample(xt, m, NCHW=False)
        outs.append(idx)
    torch.cuda.synchronize()
    _fps_form("default")
    assert sampling.furthest_sampling_status(cuda) == 0
    for idx in outs:
        assert torch.equal(idx, ref)
    assert torch.equal(furthest_point_sample(xt, m, NCHW=False)[0], ref)      # the bucketed kernel
    e_idx, _ = oracle.furthest_sampling(x[:1], m, 0)
    assert np.array_equal(ref[:1].cpu().numpy(), e_idx)


@pytest.mark.parametrize("b,n,m,seed", [(2, 2048, 256, 0), (1, 300, 64, 7), (16, 8192, 200, 5), (3, 1024, 1, 9), (1, 70000, 24, 0)])
def test_fps_writes_the_picked_points_in_the_same_launch(cuda, fps_path, b, n, m, seed):
    """furthest_point_sample = sampling + gather_points of the coordinates (reference geo_operations.py:59-63), here one
    launch: both layouts equal the composition, through both kernels, gradients included"""
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.network.geo_operations import furthest_point_sample, FurthestPointSampling
    from pytorch_points_amd.network.operations import gather_points
    x = S.unit_sphere(40 + n, b, n)
    e_idx, _ = oracle.furthest_sampling(x, m, seed)
    xt = _t(x, cuda).requires_grad_(True)
    idx, chosen = furthest_point_sample(xt, m, NCHW=False, seedIdx=seed)               # (B, m, 3)
    assert np.array_equal(idx.cpu().numpy(), e_idx)
    ref = np.take_along_axis(x, e_idx[..., None].astype(np.int64), 1)
    assert np.array_equal(chosen.detach().cpu().numpy(), ref)
    w = _t(S.normal(41, (b, m, 3)), cuda)
    (chosen * w).sum().backward()
    x2 = _t(x, cuda).requires_grad_(True)                                              # the composition
    i2 = FurthestPointSampling.apply(x2, m, seed)
    c2 = gather_points(x2.transpose(2, 1).contiguous(), i2).transpose(2, 1)
    (c2 * w).sum().backward()
    assert torch.equal(i2, idx) and torch.equal(c2.detach(), chosen.detach())
    assert torch.allclose(xt.grad, x2.grad, rtol=1e-6, atol=1e-7)
    idx_c, chosen_c = furthest_point_sample(_t(np.ascontiguousarray(x.transpose(0, 2, 1)), cuda), m, NCHW=True, seedIdx=seed)
    assert torch.equal(idx_c, idx) and np.array_equal(chosen_c.cpu().numpy(), ref.transpose(0, 2, 1))
    # the C ABI's points-last layout, and temp as an in/out argument as in the reference's call
    temp = torch.full((b, n), 1e10, device=cuda)
    out_i = torch.empty(b, m, dtype=torch.int32, device=cuda)
    out_p = torch.empty(b, m, 3, device=cuda)
    sampling.furthest_sampling(m, seed, _t(x, cuda), temp, out_i, out_p, False)
    assert np.array_equal(out_i.cpu().numpy(), e_idx) and np.array_equal(out_p.cpu().numpy(), ref)
    with pytest.raises(RuntimeError, match="sampled"):
        sampling.furthest_sampling(m, seed, _t(x, cuda), temp, out_i, torch.empty(b, m + 1, 3, device=cuda), False)


def test_fps_timeout_is_reported_per_stream(cuda):
    """a timed-out cluster wait sets the sticky status word of THAT stream's workspace; it is reported by
    furthest_sampling_check() or by the next call on the same device and stream -- not by, nor wiped out by, a clean
    call on another stream.  (The timeout itself is faked by setting the word the kernel would set.)"""
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    x = _t(S.unit_sphere(15, 16, 8192), cuda)                  # a shape served by the cluster kernel (has a workspace)
    ref, _ = furthest_point_sample(x, 64, NCHW=False)
    sampling.furthest_sampling_check(cuda)                     # clean so far
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        other, _ = furthest_point_sample(x, 64, NCHW=False)    # the side stream's own workspace and mirror
        side.synchronize()
        sampling.furthest_sampling_check(cuda)
    ws = _lib._WS[(cuda.index, _lib.raw_stream(cuda), "fps")]
    ws[:4].view(torch.int32).fill_(1)                          # "a wait of the last call on the main stream timed out"
    again, _ = furthest_point_sample(x, 64, NCHW=False)        # copies the word behind itself
    with torch.cuda.stream(side):
        furthest_point_sample(x, 64, NCHW=False)               # a clean call elsewhere neither reports nor clears it
        sampling.furthest_sampling_check(cuda)
    with pytest.raises(RuntimeError, match="timed out"):
        sampling.furthest_sampling_check(cuda)
    sampling.furthest_sampling_check(cuda)                     # reported once, then cleared (mirror and device word)
    assert sampling.furthest_sampling_status(cuda) == 0
    ws[:4].view(torch.int32).fill_(1)
    furthest_point_sample(x, 64, NCHW=False)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="timed out"):       # ... or by the next call on the same stream
        furthest_point_sample(x, 64, NCHW=False)
    idx, _ = furthest_point_sample(x, 64, NCHW=False)
    assert torch.equal(idx, ref) and torch.equal(other, ref) and torch.equal(again, ref)
    assert sampling.furthest_sampling_status(cuda) == 0


def test_gather_matches_torch_and_backward(cuda):
    from pytorch_points_amd.network.operations import gather_points
    b, c, n, m = 3, 37, 500, 123
    f = _t(S.normal(20, (b, c, n)), cuda).requires_grad_(True)
    idx = _t((S.uniform01(21, (b, m)).reshape(b, m) * n).astype(np.int32), cuda)
    out = gather_points(f, idx)
    ref = torch.gather(f, 2, idx.long()[:, None, :].expand(-1, c, -1))
    assert torch.equal(out, ref)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.gather_forward(f.detach().cpu().numpy(), idx.cpu().numpy()))
    w = _t(S.normal(22, (b, c, m)), cuda)
    (out * w).sum().backward()
    g = f.grad.clone()
    f.grad = None
    (ref * w).sum().backward()
    assert torch.allclose(g, f.grad, rtol=1e-5, atol=1e-6)
    # int64 idx is accepted and converted (reference operations.py:55)
    assert torch.equal(gather_points(f.detach(), idx.long()), out.detach())


# ----------------------------------------------------------------------------------- ball query
@pytest.fixture(params=["grid", "grid_lpc4", "grid_lpc1", "split", "single_wave"], autouse=False)
def bq_path(request, cuda):
    """the ball_query paths: uniform grid (default for N >= 2048, scan fallback) with 2 (default), 4
    or 1 lanes per centre; scan with 4 waves x cloud quarters; scan with one wave per centre tile"""
    import ctypes
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_ball_query_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    search = _lib.lib().pp_debug_set_ball_query_search
    search.argtypes = [ctypes.c_int]
    search.restype = None
    lpc = _lib.lib().pp_debug_set_ball_query_lpc
    lpc.argtypes = [ctypes.c_int]
    lpc.restype = None
    setter(1 if request.param == "single_wave" else 0)
    search(2 if request.param.startswith("grid") else 1)   # 2: grid from N = 2048 (automatic: from 4096)
    lpc({"grid_lpc4": 4, "grid_lpc1": 1}.get(request.param, 0))
    yield request.param
    setter(0)
    search(0)
    lpc(0)


@pytest.mark.parametrize("r", [0.05, 0.2, 0.5])
@pytest.mark.parametrize("ns", [16, 64])
def test_ball_query_matches_oracle(cuda, bq_path, r, ns):
    from pytorch_points_amd.network.operations import ball_query
    x = S.unit_sphere(30, 2, 2048)
    fidx, _ = oracle.furthest_sampling(x, 256, 0)
    centres = np.take_along_axis(x, fidx[..., None].astype(np.int64), 1)
    idx = ball_query(r, ns, _t(x, cuda), _t(centres, cuda))
    assert idx.dtype == torch.int32 and idx.shape == (2, 256, ns) and not idx.requires_grad
    assert np.array_equal(idx.cpu().numpy(), oracle.ball_query(centres, x, r, ns))


def test_ball_query_grid_adversarial(cuda, bq_path):
    """grid path on data built to break it: duplicates, planar / collinear clouds, centres far outside
    the cloud, tight clusters, mixed batch (one element falls back to the scan: big radius in cells)."""
    from pytorch_points_amd._ext import sampling
    n, m = 4096, 300
    cases = []
    x = S.unit_sphere(33, 1, n)
    x[0, n // 2:] = x[0, : n // 2]
    cases.append((x, x[:, ::13].copy()[:, :m], 0.12, 24))                       # duplicates
    pl = S.uniform01(34, (1, n, 3)).reshape(1, n, 3).astype(np.float32); pl[..., 2] = 0.25
    cases.append((pl, pl[:, :m].copy() + np.float32(0.01), 0.08, 16))           # planar
    ln = np.zeros((1, n, 3), np.float32); ln[0, :, 1] = np.linspace(0, 1, n, dtype=np.float32)
    cases.append((ln, ln[:, ::11].copy()[:, :m], 0.01, 32))                     # collinear
    far = S.unit_sphere(35, 1, m) * np.float32(30)
    cases.append((S.unit_sphere(36, 1, n), far, 0.5, 8))                        # empty balls far away
    cl = (S.normal(37, (1, n, 3)) * 1e-3).astype(np.float32)
    cases.append((cl, cl[:, :m].copy(), 2e-3, 64))                              # one tight cluster: dense balls
    mixed = np.concatenate([S.unit_sphere(38, 1, n), S.unit_sphere(39, 1, n) * np.float32(1e-2)], 0)
    cases.append((mixed, mixed[:, ::9].copy()[:, :m], 0.05, 20))                # element 1: radius >> cell
    off = S.unit_sphere(40, 1, n) + np.float32(2000.0)
    cases.append((off, off[:, ::5].copy()[:, :m], 0.07, 32))                    # coordinates >> radius and cell size
    for k, (x, c, r, ns) in enumerate(cases):
        x = np.ascontiguousarray(x); c = np.ascontiguousarray(c)
        got = sampling.ball_query(_t(c, cuda), _t(x, cuda), r, ns).cpu().numpy()
        want = oracle.ball_query(c, x, r, ns)
        bad = np.argwhere((got != want).any(-1))
        assert bad.size == 0, "case %d: %d rows differ, first %s got %s want %s" % (
            k, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])])


@pytest.mark.parametrize("b,n,m,r,ns", [(1, 100, 70, 0.3, 5), (2, 1000, 300, 1e-4, 8), (1, 37, 3, 10.0, 200),
                                        (1, 2000, 515, 0.25, 33), (1, 9, 1, 0.5, 1), (1, 64, 64, 0.4, 300),
                                        (1, 70000, 130, 0.05, 40), (2, 4099, 777, 0.3, 128), (1, 33, 65, 2.0, 7),
                                        (3, 16384, 1000, 0.1, 64), (2, 5000, 600, 0.02, 16), (1, 8192, 300, 0.6, 32),
                                        (2, 3000, 200, 0.15, 3), (1, 3000, 300, 0.1, 24), (1, 140000, 100, 0.05, 8), (1, 300000, 500, 0.02, 16), (1, 600000, 64, 0.02, 8),
                                        (2, 2048, 4096, 0.2, 12)])
def test_ball_query_edges(cuda, bq_path, b, n, m, r, ns):
    """odd sizes; empty balls (all-zero rows); full balls (early exit); nsample beyond the LDS
    staging limit (direct-store path)."""
    from pytorch_points_amd._ext import sampling
    x = S.unit_sphere(31, b, n)
    c = S.unit_sphere(32, b, m)
    idx = sampling.ball_query(_t(c, cuda), _t(x, cuda), r, ns)
    assert np.array_equal(idx.cpu().numpy(), oracle.ball_query(c, x, r, ns))


# --------------------------------------------------------------------------------- group points
@pytest.fixture(params=["auto", "global_atomics", "lds_columns", "lds_columns_f32"])
def group_grad_path(request, cuda):
    """auto = sorted-triples scatter-add where it qualifies (scatter.hip); the others switch it off and
    force global atomics, the LDS column in double (ds_add_f64) or the LDS column in fp32."""
    import ctypes
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_group_points_grad_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    smode = _lib.lib().pp_debug_set_scatter_mode
    smode.argtypes = [ctypes.c_int]
    smode.restype = None
    setter({"auto": 0, "global_atomics": 1, "lds_columns": 2, "lds_columns_f32": 3}[request.param])
    smode(0 if request.param == "auto" else 1)
    yield request.param
    setter(0)
    smode(0)


@pytest.fixture(params=["sorted", "atomics"])
def scatter_path(request, cuda):
    import ctypes
    from pytorch_points_amd import _lib
    smode = _lib.lib().pp_debug_set_scatter_mode
    smode.argtypes = [ctypes.c_int]
    smode.restype = None
    smode(0 if request.param == "sorted" else 1)
    yield request.param
    smode(0)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("b,c,n,m", [(9, 64, 16384, 4096), (4, 130, 4096, 1024), (17, 33, 8200, 8192), (2, 300, 1024, 2052),
                                     (2, 256, 16380, 1028), (9, 64, 16383, 4096), (5, 120, 4097, 1025), (3, 200, 30000, 2048)])
def test_gather_forward_large(cuda, variant, b, c, n, m):
    """shapes the LDS-staged gather takes (variant 0) and the same through the global-gather kernel"""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import sampling
    f = _t(S.normal(28, (b, c, n)), cuda)
    idx = _t((S.uniform01(29, (b, m)).reshape(b, m) * n).astype(np.int32), cuda)
    idx[:, 0] = n - 1
    idx[:, -1] = 0
    out = torch.empty(b, c, m, device=cuda)
    setter = _lib.lib().pp_debug_set_gather_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(variant)
    try:
        sampling.gather_forward(b, c, n, m, f, idx, out)
    finally:
        setter(0)
    assert torch.equal(out, torch.gather(f, 2, idx.long()[:, None, :].expand(-1, c, -1)))


@pytest.mark.parametrize("b,c,n,m", [(3, 40, 5000, 20001), (9, 16, 16384, 8192), (1, 130, 777, 40000)])
def test_gather_backward_large(cuda, scatter_path, b, c, n, m):
    from pytorch_points_amd._ext import sampling
    go = _t(S.normal(23, (b, c, m)), cuda)
    idx = _t((S.uniform01(24, (b, m)).reshape(b, m) * n).astype(np.int32), cuda)
    gp = torch.zeros(b, c, n, device=cuda)
    sampling.gather_backward(b, c, n, m, go, idx, gp)
    ref = torch.zeros(b, c, n, device=cuda, dtype=torch.float64)
    ref.scatter_add_(2, idx.long()[:, None, :].expand(-1, c, -1), go.double())
    assert torch.allclose(gp.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.fixture(params=["auto", "global_atomics", "lds_columns", "sorted"])
def interp_grad_path(request, cuda):
    """three_interpolate_grad forms: global atomics (the reference's), LDS columns in double
    (ds_add_f64), sorted triples (scatter.hip)"""
    import ctypes
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_three_interpolate_grad_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    smode = _lib.lib().pp_debug_set_scatter_mode
    smode.argtypes = [ctypes.c_int]
    smode.restype = None
    setter({"auto": 0, "global_atomics": 1, "lds_columns": 2, "sorted": 3}[request.param])
    smode(1 if request.param == "global_atomics" else 0)
    yield request.param
    setter(0)
    smode(0)


@pytest.mark.parametrize("b,c,n,m", [(2, 33, 16384, 4096), (9, 8, 5001, 700), (1, 64, 40000, 20480), (3, 6, 3000, 9000),
                                     (2, 3, 2500, 19456)])
def test_three_interpolate_grad_large(cuda, interp_grad_path, b, c, n, m):
    from pytorch_points_amd._ext import sampling
    go = _t(S.normal(25, (b, c, n)), cuda)
    idx = _t((S.uniform01(26, (b, n, 3)).reshape(b, n, 3) * m).astype(np.int32), cuda)
    w = _t(S.uniform01(27, (b, n, 3)).reshape(b, n, 3).astype(np.float32), cuda)
    gp = torch.zeros(b, c, m, device=cuda)
    sampling.three_interpolate_grad_wrapper(b, c, n, m, go, idx, w, gp)
    ref = torch.zeros(b, c, m, device=cuda, dtype=torch.float64)
    contrib = go.double()[:, :, :, None] * w.double()[:, None]                       # (b,c,n,3)
    ref.scatter_add_(2, idx.long()[:, None].expand(-1, c, -1, -1).reshape(b, c, -1), contrib.reshape(b, c, -1))
    assert torch.allclose(gp.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,c,n,npoint,ns", [(2, 8, 2048, 256, 16), (1, 3, 100, 7, 5), (2, 67, 500, 33, 12), (1, 1, 10, 1, 1),
                                             (9, 4, 4096, 512, 32), (4, 16, 4096, 1024, 32), (3, 10, 5000, 700, 48),
                                             # shapes without 16-byte alignment / rows beyond 64 KiB (4-byte LDS-staged forms)
                                             (9, 5, 16383, 1024, 64), (2, 6, 20000, 4096, 64), (4, 7, 4096, 4095, 33),
                                             (2, 5, 3000, 1023, 33), (5, 4, 30001, 999, 107)])
def test_group_points_matches_torch_and_backward(cuda, group_grad_path, b, c, n, npoint, ns):
    from pytorch_points_amd.network.operations import grouping_operation
    f = _t(S.normal(40, (b, c, n)), cuda).requires_grad_(True)
    idx = _t((S.uniform01(41, (b, npoint, ns)).reshape(b, npoint, ns) * n).astype(np.int32), cuda)
    out = grouping_operation(f, idx)
    ref = torch.gather(f, 2, idx.long().reshape(b, 1, -1).expand(-1, c, -1)).reshape(b, c, npoint, ns)
    assert out.shape == (b, c, npoint, ns) and torch.equal(out, ref)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.group_points(f.detach().cpu().numpy(), idx.cpu().numpy()))
    w = _t(S.normal(42, (b, c, npoint, ns)), cuda)
    (out * w).sum().backward()
    g = f.grad.clone()
    f.grad = None
    (ref * w).sum().backward()
    assert torch.allclose(g, f.grad, rtol=1e-5, atol=1e-5)
    e = oracle.group_points_grad(w.cpu().numpy(), idx.cpu().numpy(), n)
    assert np.allclose(g.cpu().numpy(), e, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,c,n,npoint,ns,r", [(2, 6, 4096, 512, 64, 0.08), (1, 5, 50000, 1024, 32, 0.02),
                                               (9, 3, 20001, 256, 16, 0.05), (2, 4, 2048, 300, 24, 0.6)])
def test_group_points_grad_ball_rows_and_split_columns(cuda, group_grad_path, b, c, n, npoint, ns, r):
    """index rows as ball_query writes them (ascending hits, then the first one repeated: runs of equal
    consecutive indices, merged in registers by the double-column kernel), and clouds too large for
    one LDS column (n > 19456: the column is split into destination ranges)."""
    from pytorch_points_amd._ext import sampling
    x = S.unit_sphere(45, b, n)
    idx = sampling.ball_query(_t(np.ascontiguousarray(x[:, :npoint]), cuda), _t(x, cuda), r, ns)
    idx[:, 0, :] = n - 1                 # a whole row on the last destination
    idx[:, -1, 1::2] = idx[:, -1, 0:1]   # alternating: no runs longer than one
    go = _t(S.normal(46, (b, c, npoint, ns)), cuda)
    got = sampling.group_points_grad(go, idx, n)
    ref = torch.zeros(b, c, n, device=cuda, dtype=torch.float64)
    ref.scatter_add_(2, idx.long().reshape(b, 1, -1).expand(-1, c, -1), go.double().reshape(b, c, -1))
    # destinations here collect up to a few hundred addends: the forms that accumulate in fp32 in
    # arbitrary order (the reference's global atomics, the fp32 LDS column) carry that many roundings
    atol = 1e-4 if group_grad_path in ("global_atomics", "lds_columns_f32") else 1e-5
    assert torch.allclose(got.double(), ref, rtol=1e-5, atol=atol)
    e = oracle.group_points_grad(go.cpu().numpy(), idx.cpu().numpy(), n)
    assert np.allclose(got.cpu().numpy(), e, rtol=1e-5, atol=1e-4)   # the oracle sums in fp32 too


@pytest.mark.parametrize("b,c,n,npoint,ns", [(2, 6, 4096, 512, 64), (3, 5, 1000, 77, 9), (1, 4, 30000, 2048, 32), (2, 3, 600, 10, 4),
                                             (2, 8, 16384, 1875, 16), (1, 16, 8192, 4097, 20)])
def test_group_points_grad_accumulating_and_overwriting_abi(cuda, b, c, n, npoint, ns):
    """the two contracts of the C ABI side by side: pp_group_points_grad_ws_f32 ACCUMULATES into the caller's tensor (the
    reference's: a zero-filled output it adds into, _ext/sampling.cpp:148-150), pp_group_points_grad_out_ws_f32 WRITES every
    element (what the shim calls on uninitialised memory since round 5) -- on every launch path the sizes select"""
    from pytorch_points_amd import _lib
    idx = _t((S.uniform01(47, (b, npoint, ns)).reshape(b, npoint, ns) * n).astype(np.int32), cuda)
    idx[:, 0, :] = n - 1                       # a run across a whole row
    idx[:, -3:, :] = 5                         # ... and one over the last rows: where P / 4 is no multiple of 64
    go = _t(S.normal(48, (b, c, npoint, ns)), cuda)   # the wave that straddles the end takes the lane-by-lane loop
    ref = torch.zeros(b, c, n, device=cuda, dtype=torch.float64)
    ref.scatter_add_(2, idx.long().reshape(b, 1, -1).expand(-1, c, -1), go.double().reshape(b, c, -1))
    L = _lib.lib()
    acc = torch.full((b, c, n), 3.0, device=cuda)
    out = torch.full((b, c, n), float("nan"), device=cuda)       # must be overwritten everywhere
    with _lib.on_device(cuda) as stream:
        args = lambda t: (_lib.ptr(go), _lib.ptr(idx), _lib.ptr(t), b, c, n, npoint, ns, c * npoint * ns, None, 0, stream)
        _lib.check(L.pp_group_points_grad_ws_f32(*args(acc)), "accumulate")
        _lib.check(L.pp_group_points_grad_out_ws_f32(*args(out)), "overwrite")
    torch.cuda.synchronize()
    assert torch.allclose(acc.double(), ref + 3.0, rtol=1e-5, atol=1e-4)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("variant", [0, 1, 2, 4, 8, 516, 604, 608, 616])
@pytest.mark.parametrize("b,c,n,npoint,ns", [(9, 5, 1024, 600, 16), (2, 4, 16384, 2048, 64), (1, 7, 500, 4099, 32),
                                             (3, 9, 10000, 1024, 64), (2, 6, 20480, 512, 64), (2, 13, 4096, 2048, 32), (1, 16, 65536, 4096, 32),
                                             (4, 16, 16384, 4096, 64)])
def test_group_points_every_kernel_variant(cuda, variant, b, c, n, npoint, ns):
    """global-gather kernel and the LDS-staged kernel (2/4/8 index quads per thread) agree with
    torch.gather, including ragged tails and batch counts that are not a multiple of 8."""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import sampling
    f = _t(S.normal(43, (b, c, n)), cuda)
    idx = _t((S.uniform01(44, (b, npoint, ns)).reshape(b, npoint, ns) * n).astype(np.int32), cuda)
    idx[:, 0, 0] = n - 1
    idx[:, -1, -1] = 0
    setter = _lib.lib().pp_debug_set_group_points_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(variant)
    try:
        out = sampling.group_points(f, idx)
    finally:
        setter(0)
    ref = torch.gather(f, 2, idx.long().reshape(b, 1, -1).expand(-1, c, -1)).reshape(b, c, npoint, ns)
    assert torch.equal(out, ref)


def test_group_points_preconditions(cuda):
    from pytorch_points_amd._ext import sampling
    f = torch.zeros(1, 2, 8, device=cuda)
    idx = torch.zeros(1, 2, 2, dtype=torch.int32, device=cuda)
    with pytest.raises(RuntimeError):
        sampling.group_points(f, idx.long())            # CHECK_IS_INT
    with pytest.raises(RuntimeError):
        sampling.group_points(f.transpose(1, 2), idx)   # CHECK_CONTIGUOUS
    with pytest.raises(RuntimeError, match="CPU not supported"):
        sampling.group_points(f.cpu(), idx.cpu())


def test_query_and_group_composed(cuda):
    from pytorch_points_amd.network.operations import QueryAndGroup
    from pytorch_points_amd.network import pointnet2_utils
    b, n, npoint, c, r, ns = 2, 1024, 128, 6, 0.3, 16
    x = S.unit_sphere(50, b, n)
    centres = x[:, ::8].copy()
    feats = S.normal(51, (b, c, n))
    qg = QueryAndGroup(r, ns, use_xyz=True)
    out = qg(_t(x, cuda), _t(centres, cuda), _t(feats, cuda))
    idx = oracle.ball_query(centres, x, r, ns)
    gx = oracle.group_points(np.ascontiguousarray(x.transpose(0, 2, 1)), idx) - centres.transpose(0, 2, 1)[..., None]
    gf = oracle.group_points(feats, idx)
    exp = np.concatenate([gx, gf], 1)
    assert out.shape == (b, 3 + c, npoint, ns)
    assert np.array_equal(out.cpu().numpy(), exp)
    assert pointnet2_utils.QueryAndGroup is QueryAndGroup
    out2 = QueryAndGroup(r, ns, use_xyz=True)(_t(x, cuda), _t(centres, cuda), None)
    assert np.array_equal(out2.cpu().numpy(), gx)
    ga = pointnet2_utils.GroupAll()(_t(x, cuda), None, _t(feats, cuda))
    assert ga.shape == (b, 3 + c, 1, n)


@pytest.mark.parametrize("use_xyz,with_features", [(True, True), (True, False), (False, True)])
def test_query_and_group_fused_equals_composition(cuda, use_xyz, with_features):
    """The fused caller (one output tensor, no torch.cat) == the reference's op-by-op composition,
    forward bitwise and backward for xyz, new_xyz and features."""
    from pytorch_points_amd.network.operations import QueryAndGroup
    b, n, npoint, c, r, ns = 3, 2048, 256, 9, 0.25, 32
    x = S.unit_sphere(52, b, n)
    qg = QueryAndGroup(r, ns, use_xyz=use_xyz)

    def run(fn):
        xyz = _t(x, cuda).requires_grad_(True)
        new_xyz = _t(x[:, ::8].copy(), cuda).requires_grad_(True)
        feats = _t(S.normal(53, (b, c, n)), cuda).requires_grad_(True) if with_features else None
        out = fn(xyz, new_xyz, feats)
        w = _t(S.normal(54, tuple(out.shape)), cuda)
        (out * w).sum().backward()
        return out.detach(), xyz.grad, new_xyz.grad, (feats.grad if with_features else None)

    fused = run(qg.forward)
    comp = run(qg.forward_unfused)
    assert torch.equal(fused[0], comp[0])
    assert fused[0].shape == (b, (3 if use_xyz else 0) + (c if with_features else 0), npoint, ns)
    for g, e in zip(fused[1:], comp[1:]):
        if e is None:
            assert g is None or float(g.abs().sum()) == 0
        else:
            assert torch.allclose(g, e, rtol=1e-4, atol=1e-4)


# --------------------------------------------------------------------- three_nn / three_interpolate
@pytest.mark.parametrize("b,n,m", [(2, 2048, 256), (1, 10, 2), (1, 300, 3), (1, 5, 1), (2, 1000, 1000)])
def test_three_nn_matches_oracle(cuda, b, n, m):
    from pytorch_points_amd.network.pointnet2_utils import three_nn
    u = S.unit_sphere(60, b, n)
    k = S.unit_sphere(61, b, m)
    if m >= 6:
        k[:, m // 2:m // 2 + 3] = k[:, :3]          # exact ties: the earlier index wins at every rank
    dist, idx = three_nn(_t(u, cuda), _t(k, cuda))
    e_d2, e_idx = oracle.three_nn(u, k)
    assert idx.dtype == torch.int32 and np.array_equal(idx.cpu().numpy(), e_idx)
    with np.errstate(invalid="ignore"):
        assert np.array_equal(dist.cpu().numpy(), np.sqrt(e_d2))     # wrapper returns sqrt (ref :33)
    if m < 3:
        assert np.isinf(dist.cpu().numpy()[..., m:]).all() and (e_idx[..., m:] == 0).all()


@pytest.fixture(params=["grid", "scan"])
def tn_path(request, cuda):
    import ctypes
    from pytorch_points_amd import _lib
    search = _lib.lib().pp_debug_set_three_nn_search
    search.argtypes = [ctypes.c_int]
    search.restype = None
    search(0 if request.param == "grid" else 1)
    yield request.param
    search(0)


@pytest.mark.parametrize("b,n,m", [(2, 4096, 1024), (3, 5000, 3001), (1, 16384, 4096), (2, 1024, 8192), (1, 70000, 1500)])
def test_three_nn_grid_matches_oracle(cuda, tn_path, b, n, m):
    from pytorch_points_amd._ext import sampling
    u = S.unit_sphere(62, b, n)
    k = S.unit_sphere(63, b, m)
    k[:, m // 2:m // 2 + 5] = k[:, :5]                # exact ties at every rank
    u[:, :7] = k[:, 10:17]                            # zero distances
    d2 = torch.empty(b, n, 3, device=cuda)
    idx = torch.empty(b, n, 3, dtype=torch.int32, device=cuda)
    sampling.three_nn_wrapper(b, n, m, _t(u, cuda), _t(k, cuda), d2, idx)
    e_d2, e_idx = oracle.three_nn(u, k)
    assert np.array_equal(idx.cpu().numpy(), e_idx) and np.array_equal(d2.cpu().numpy(), e_d2)


def test_three_nn_grid_adversarial(cuda, tn_path):
    """data built to break the grid search: cloud far from the origin (coordinates >> cell size),
    planar / collinear / clustered known points, unknown points far outside the known cloud, all
    known points identical (useless grid -> scan fallback for that batch element only), duplicates"""
    from pytorch_points_amd._ext import sampling
    n, m = 3000, 2048
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
    dup = S.unit_sphere(75, 1, m); dup[0, m // 4:] = dup[0, : m - m // 4]
    cases.append((dup[:, ::-1][:, :n // 2].repeat(2, 1).copy(), dup))
    for i, (u, k) in enumerate(cases):
        u = np.ascontiguousarray(u, np.float32); k = np.ascontiguousarray(k, np.float32)
        b = u.shape[0]
        d2 = torch.empty(b, u.shape[1], 3, device=cuda)
        idx = torch.empty(b, u.shape[1], 3, dtype=torch.int32, device=cuda)
        sampling.three_nn_wrapper(b, u.shape[1], k.shape[1], _t(u, cuda), _t(k, cuda), d2, idx)
        e_d2, e_idx = oracle.three_nn(u, k)
        bad = np.argwhere((idx.cpu().numpy() != e_idx).any(-1))
        assert bad.size == 0, "case %d: %d rows differ, first %s got %s want %s" % (
            i, len(bad), bad[0], idx.cpu().numpy()[tuple(bad[0])], e_idx[tuple(bad[0])])
        assert np.array_equal(d2.cpu().numpy(), e_d2), "case %d distances" % i


def _multi_round_cases():
    """known / reference clouds on which a query's first round (a box of about one cell around it) finds fewer than the
    neighbours it needs, so that the later rounds -- which walk only the BALL of the farthest neighbour found so far,
    rows beyond it passed over, the others cut along x (round 5; ADVICE r5: untested) -- decide the answer: sparse
    clouds, queries beside and far outside the known box, outliers (the grid trimmed to the bulk: rim cells open
    outwards), clouds far from the origin, a handful of known points"""
    rng = np.random.default_rng(91)
    cases = []
    sparse = (rng.random((1, 300, 3), dtype=np.float32) * np.float32(40.0))
    cases.append(("sparse", (rng.random((1, 2500, 3), dtype=np.float32) * np.float32(40.0)), sparse))
    beside = S.unit_sphere(92, 1, 2048)
    q = S.unit_sphere(93, 1, 3000) * np.float32(0.3)
    q[..., 0] += np.float32(4.0)                         # queries beside the known box, a few boxes away
    q[0, ::7] *= np.float32(25.0)                        # ... and some far outside in every direction
    cases.append(("beside_and_far", q, beside))
    outl = S.unit_sphere(94, 1, 4096).copy()
    outl[0, ::500] *= np.float32(300.0)                  # a few outliers: the box is trimmed to the bulk
    qo = np.concatenate([S.unit_sphere(95, 1, 1500), outl[:, ::500] + np.float32(0.5), S.unit_sphere(96, 1, 500) * np.float32(120.0)], 1)
    cases.append(("outliers_trimmed", qo, outl))
    off = S.unit_sphere(97, 1, 700) * np.float32(3.0) + np.float32(1000.0)
    cases.append(("offset_1e3_sparse", S.unit_sphere(98, 1, 2000) * np.float32(5.0) + np.float32(1000.0), off))
    few = (rng.random((1, 5, 3), dtype=np.float32))
    cases.append(("five_known", S.unit_sphere(99, 1, 1000) * np.float32(2.0), few))
    return [(n, np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)) for n, a, b in cases]


def test_three_nn_and_knn_multi_round_searches(cuda):
    from pytorch_points_amd._ext import sampling
    from pytorch_points_amd.ops import knn_points
    for name, q, k in _multi_round_cases():
        b, n, m = q.shape[0], q.shape[1], k.shape[1]
        d2 = torch.empty(b, n, 3, device=cuda)
        idx = torch.empty(b, n, 3, dtype=torch.int32, device=cuda)
        sampling.three_nn_wrapper(b, n, m, _t(q, cuda), _t(k, cuda), d2, idx)
        e_d2, e_idx = oracle.three_nn(q, k)
        assert np.array_equal(idx.cpu().numpy(), e_idx), "three_nn %s" % name
        assert np.array_equal(d2.cpu().numpy(), e_d2), "three_nn %s distances" % name
        for K in (1, 4, 17, min(m, 64), m):             # (K near M: the last rounds take the whole grid)
            if K > 128:
                continue
            out = knn_points(_t(q, cuda), _t(k, cuda), K=K)
            e_d, e_i = oracle.knn(q, k, K)
            assert np.array_equal(out.idx.cpu().numpy(), e_i), "knn %s K=%d" % (name, K)
            assert np.array_equal(out.dists.cpu().numpy(), e_d), "knn %s K=%d distances" % (name, K)


def test_three_interpolate_forward_backward(cuda):
    from pytorch_points_amd.network.pointnet2_utils import three_interpolate
    b, c, m, n = 2, 19, 256, 777
    feats = _t(S.normal(70, (b, c, m)), cuda).requires_grad_(True)
    idx = _t((S.uniform01(71, (b, n, 3)).reshape(b, n, 3) * m).astype(np.int32), cuda)
    w = S.uniform01(72, (b, n, 3)).reshape(b, n, 3).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    wt = _t(w, cuda)
    out = three_interpolate(feats, idx, wt)
    e = oracle.three_interpolate(feats.detach().cpu().numpy(), idx.cpu().numpy(), w)
    assert np.array_equal(out.detach().cpu().numpy(), e)             # canonical fma order: bit-exact
    g = torch.gather(feats.unsqueeze(2).expand(-1, -1, n, -1), 3, idx.long()[:, None].expand(-1, c, -1, -1))
    ref = (g * wt[:, None]).sum(-1)
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    go = _t(S.normal(73, (b, c, n)), cuda)
    (out * go).sum().backward()
    gr = feats.grad.clone()
    feats.grad = None
    (ref * go).sum().backward()
    assert torch.allclose(gr, feats.grad, rtol=1e-5, atol=1e-5)
    eg = oracle.three_interpolate_grad(go.cpu().numpy(), idx.cpu().numpy(), w, m)
    assert np.allclose(gr.cpu().numpy(), eg, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("b,c,m,n", [(9, 16, 4096, 16384), (2, 7, 1000, 70000), (3, 5, 16384, 40000), (1, 64, 512, 131072),
                                     (17, 130, 2048, 4096), (32, 67, 4096, 8192), (9, 40, 4095, 16383), (8, 36, 4099, 16384),
                                     (3, 90, 20000, 9000), (5, 70, 30001, 4097)])
def test_three_interpolate_both_kernels(cuda, variant, b, c, m, n):
    """channel-group (0, where it applies), global-gather (1) and row-at-a-time LDS (2) forms == oracle
    bitwise (canonical fma order)."""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import sampling
    feats = _t(S.normal(74, (b, c, m)), cuda)
    idx = _t((S.uniform01(75, (b, n, 3)).reshape(b, n, 3) * m).astype(np.int32), cuda)
    idx[:, 0, 0] = m - 1
    w = _t(S.uniform01(76, (b, n, 3)).reshape(b, n, 3).astype(np.float32), cuda)
    out = torch.empty(b, c, n, device=cuda)
    setter = _lib.lib().pp_debug_set_three_interpolate_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(variant)
    try:
        sampling.three_interpolate_wrapper(b, c, m, n, feats, idx, w, out)
    finally:
        setter(0)
    e = oracle.three_interpolate(feats[:2].cpu().numpy(), idx[:2].cpu().numpy(), w[:2].cpu().numpy())
    assert np.array_equal(out[:2].cpu().numpy(), e)
    g = torch.gather(feats.unsqueeze(2).expand(-1, -1, n, -1), 3, idx.long()[:, None].expand(-1, c, -1, -1))
    assert torch.allclose(out, (g * w[:, None]).sum(-1), rtol=1e-5, atol=1e-5)


def test_fp_module_style_pipeline(cuda):
    """three_nn -> inverse-distance weights -> three_interpolate, as PointnetFPModule does
    (reference network/pointnet2_modules.py:136-141)."""
    from pytorch_points_amd.network.pointnet2_utils import three_nn, three_interpolate
    b, n, m, c = 2, 600, 150, 8
    unknown = _t(S.unit_sphere(80, b, n), cuda)
    known = _t(S.unit_sphere(81, b, m), cuda)
    kf = _t(S.normal(82, (b, c, m)), cuda)
    dist, idx = three_nn(unknown, known)
    dist_recip = 1.0 / (dist + 1e-8)
    weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
    out = three_interpolate(kf, idx, weight)
    assert out.shape == (b, c, n) and torch.isfinite(out).all()
    e = oracle.three_interpolate(kf.cpu().numpy(), idx.cpu().numpy(), weight.cpu().numpy())
    assert np.array_equal(out.cpu().numpy(), e)


def test_config4_full_size_properties(cuda):
    """BASELINE config 4 at full size (ball_query r=0.1 nsample=64 + group_points, B=32 N=16384 C=128,
    npoint=4096): size-independent properties instead of the CPU oracle (too slow at this size) --
    every listed index is in radius, rows ascend up to the pad, the pad repeats the first hit, the row
    count equals the number of points in radius (capped), grid == scan kernels bit for bit, two batch
    elements equal the oracle; group_points equals torch.gather, its gradient equals scatter_add."""
    import ctypes
    from pytorch_points_amd import _lib
    from pytorch_points_amd._ext import sampling
    B, N, C, ns, r = 32, 16384, 128, 64, 0.1
    x = _t(S.unit_sphere(0, B, N), cuda)
    centres = x[:, ::4].contiguous()
    npoint = centres.shape[1]
    idx = sampling.ball_query(centres, x, r, ns)
    search = _lib.lib().pp_debug_set_ball_query_search
    search.argtypes = [ctypes.c_int]
    search.restype = None
    search(1)
    try:
        assert torch.equal(idx, sampling.ball_query(centres, x, r, ns))            # grid == scan
    finally:
        search(0)
    e = oracle.ball_query(centres[:2].cpu().numpy(), x[:2].cpu().numpy(), r, ns)
    assert np.array_equal(idx[:2].cpu().numpy(), e)
    li = idx.long()
    pts = torch.gather(x.unsqueeze(1).expand(-1, npoint, -1, -1), 2, li.unsqueeze(-1).expand(-1, -1, -1, 3))
    d = pts - centres.unsqueeze(2)
    d2 = torch.addcmul(torch.addcmul(d[..., 1] * d[..., 1], d[..., 0], d[..., 0]), d[..., 2], d[..., 2])
    assert bool((d2 < r * r * (1 + 1e-5)).all())                                   # every listed index is in radius
    inc = li[..., 1:] > li[..., :-1]
    cnt = 1 + inc.to(torch.int32).cumprod(-1).sum(-1)                              # length of the ascending prefix
    slot = torch.arange(ns, device=cuda)[None, None, :]
    assert bool(((slot < cnt[..., None]) | (li == li[..., :1])).all())             # beyond it: the first hit repeated
    for b0 in range(0, B, 8):                                                       # true ball population, by chunks
        dd = torch.cdist(centres[b0:b0 + 8].double(), x[b0:b0 + 8].double()) ** 2
        inside = (dd < (r * r) * (1 - 1e-6)).sum(-1)
        maybe = (dd < (r * r) * (1 + 1e-6)).sum(-1)
        c = cnt[b0:b0 + 8]
        assert bool(((c >= inside.clamp(max=ns)) & (c <= maybe.clamp(max=ns))).all())
    feats = _t(S.normal(2, (B, C, N)), cuda)
    out = sampling.group_points(feats, idx)
    flat = li.reshape(B, 1, -1)
    for c0 in range(0, C, 32):                                                      # 1 GiB at a time
        ref = torch.gather(feats[:, c0:c0 + 32], 2, flat.expand(-1, 32, -1)).reshape(B, 32, npoint, ns)
        assert torch.equal(out[:, c0:c0 + 32], ref)
        del ref
    grad = sampling.group_points_grad(out, idx, N)                                  # d/dfeats of 0.5 * |out|^2
    ref = torch.zeros(B, C, N, device=cuda, dtype=torch.float64)
    for c0 in range(0, C, 16):
        ref[:, c0:c0 + 16].scatter_add_(2, flat.expand(-1, 16, -1), out[:, c0:c0 + 16].double().reshape(B, 16, -1))
    assert torch.allclose(grad.double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,m", [(300, 40), (4096, 64), (20000, 100)])
def test_fps_without_a_temp_of_the_callers(cuda, n, m):
    """temp=None: "every point starts at 1e10 and nothing is kept" -- what the reference's wrapper does with a temp of its
    own that nobody reads back (network/geo_operations.py:32-33).  Served by the bucketed kernel where that runs; where it
    does not (small clouds: PP_ENOTSUP from the C ABI) the shim passes a buffer itself.  Same picks either way."""
    from pytorch_points_amd._ext import sampling
    x = S.unit_sphere(50 + n, 2, n)
    e_idx, _ = oracle.furthest_sampling(x, m, 3)
    idx = torch.empty(2, m, dtype=torch.int32, device=cuda)
    pts = torch.empty(2, 3, m, device=cuda)
    sampling.furthest_sampling(m, 3, _t(x, cuda), None, idx, pts, True)
    assert np.array_equal(idx.cpu().numpy(), e_idx)
    assert np.array_equal(pts.cpu().numpy(), np.take_along_axis(x, e_idx[..., None].astype(np.int64), 1).transpose(0, 2, 1))
    with pytest.raises(RuntimeError):
        sampling.furthest_sampling(m, 3, _t(x, cuda), None, idx[:, : m - 1].contiguous())
