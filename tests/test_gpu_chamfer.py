"""GPU parity: HIP Chamfer (through the autograd API -> _ext shim -> C ABI) vs the CPU oracle.

Bar (BASELINE.json north_star): indices bit-exact, distances within 1e-5 -- in fact the canonical
arithmetic makes the distances bit-exact too, and that is what is asserted."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import bruteforce as bf
from pytorch_points_amd import synthetic as S

pytestmark = pytest.mark.gpu

DIST_TOL = 1e-5  # the stated fp32 tolerance; the assertions below are stricter (bitwise)


def _clouds(b, n, m, c, dup=False):
    x1 = S.unit_sphere(0, b, n, c)
    x2 = S.unit_sphere(1, b, m, c)
    if dup:  # duplicated points: exact ties must resolve to the lowest index
        x2[:, m // 2:] = x2[:, : m - m // 2]
        x1[:, ::7] = x2[:, : len(range(0, n, 7))] if m >= len(range(0, n, 7)) else x1[:, ::7]
    return x1, x2


def _run(cuda, x1, x2):
    from pytorch_points_amd.network.model_loss import nndistance
    t1 = torch.from_numpy(x1).to(cuda)
    t2 = torch.from_numpy(x2).to(cuda)
    d1, d2, i1, i2 = nndistance(t1, t2)
    torch.cuda.synchronize()
    return d1.cpu().numpy(), i1.cpu().numpy(), d2.cpu().numpy(), i2.cpu().numpy()


SHAPES = [(2, 1024, 1024, 3), (1, 1000, 777, 3), (2, 300, 1500, 3), (1, 64, 64, 2), (1, 513, 511, 5),
          (3, 7, 5, 3), (1, 1, 1, 3), (2, 2050, 33, 3), (1, 129, 4099, 4), (1, 40, 50, 1),
          (2, 300, 257, 6), (1, 100, 333, 7), (1, 257, 100, 8), (2, 64, 200, 9), (1, 130, 70, 16), (1, 50, 90, 17)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dup", [False, True])
def test_forward_matches_oracle(cuda, shape, dup):
    b, n, m, c = shape
    x1, x2 = _clouds(b, n, m, c, dup)
    got = _run(cuda, x1, x2)
    exp = oracle.chamfer_forward(x1, x2)
    for g, e, name in zip(got, exp, ["dist1", "idx1", "dist2", "idx2"]):
        assert g.dtype == e.dtype and g.shape == e.shape
        assert np.array_equal(g, e), "%s differs at %d places" % (name, int((g != e).sum()))


@pytest.mark.parametrize("variant", [1, 2, 4, 8, 416, 216, 44, 1002, 1004, 1008, 1416, 1816, 2002, 2004, 2008, 3004])
@pytest.mark.parametrize("shape", [(2, 1024, 1024, 3), (1, 1000, 777, 3), (2, 300, 1500, 3), (1, 5, 3, 3)])
def test_forward_every_kernel_variant(cuda, variant, shape):
    """Every (queries-per-lane, group) instantiation of the C==3 kernel gives the same bits."""
    from pytorch_points_amd import _lib
    b, n, m, c = shape
    x1, x2 = _clouds(b, n, m, c, dup=True)
    exp = oracle.chamfer_forward(x1, x2)
    setter = _lib.lib().pp_debug_set_nmdistance_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(variant)
    try:
        got = _run(cuda, x1, x2)
    finally:
        setter(0)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


def test_forward_api_contract(cuda):
    from pytorch_points_amd.network.model_loss import nndistance
    x1 = torch.from_numpy(S.unit_sphere(0, 2, 100)).to(cuda).requires_grad_(True)
    x2 = torch.from_numpy(S.unit_sphere(1, 2, 90)).to(cuda).requires_grad_(True)
    d1, d2, i1, i2 = nndistance(x1, x2)
    assert d1.shape == (2, 100) and d2.shape == (2, 90) and i1.shape == (2, 100) and i2.shape == (2, 90)
    assert d1.dtype == torch.float32 and i1.dtype == torch.int32 and i2.dtype == torch.int32
    assert d1.device == x1.device and i1.device == x1.device
    assert d1.requires_grad and d2.requires_grad and not i1.requires_grad and not i2.requires_grad
    # non-contiguous input is made contiguous (reference model_loss.py:405-406)
    xt = torch.from_numpy(S.unit_sphere(0, 2, 100)).to(cuda).transpose(1, 2).contiguous().transpose(1, 2)
    assert not xt.is_contiguous()
    e1, _, j1, _ = nndistance(xt, x2.detach())
    assert torch.equal(e1, d1.detach()) and torch.equal(j1, i1)


def test_forward_empty(cuda):
    from pytorch_points_amd.network.model_loss import nndistance
    x1 = torch.zeros(2, 0, 3, device=cuda)
    x2 = torch.from_numpy(S.unit_sphere(1, 2, 9)).to(cuda)
    d1, d2, i1, i2 = nndistance(x1, x2)
    assert d1.shape == (2, 0) and d2.shape == (2, 9)
    assert float(d2.abs().sum()) == 0 and int(i2.abs().sum()) == 0  # the wrapper's zeros survive
    e = oracle.chamfer_forward(np.zeros((2, 0, 3), np.float32), x2.cpu().numpy())
    assert np.array_equal(e[2], d2.cpu().numpy())


@pytest.fixture(params=["auto", "global_atomics", "lds_columns", "csr_lists"])  # auto = double LDS accumulators for C == 3
def bwd_path(request, cuda):
    from pytorch_points_amd import _lib
    setter = _lib.lib().pp_debug_set_nmdistance_backward_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter({"auto": 0, "global_atomics": 1, "lds_columns": 2, "csr_lists": 3}[request.param])
    yield request.param
    setter(0)


@pytest.mark.parametrize("shape", [(2, 1024, 1024, 3), (1, 1000, 777, 3), (1, 64, 64, 2), (1, 513, 511, 5),
                                   (12, 2048, 3000, 3), (2, 5000, 1200, 3), (1, 4097, 4099, 3), (2, 300, 257, 6), (1, 130, 70, 16)])
def test_backward_matches_oracle_and_fp64(cuda, bwd_path, shape):
    from pytorch_points_amd.network.model_loss import nndistance
    b, n, m, c = shape
    x1, x2 = _clouds(b, n, m, c)
    t1 = torch.from_numpy(x1).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(x2).to(cuda).requires_grad_(True)
    d1, d2, i1, i2 = nndistance(t1, t2)
    w1 = torch.from_numpy(S.normal(5, (b, n))).to(cuda)
    w2 = torch.from_numpy(S.normal(6, (b, m))).to(cuda)
    ((d1 * w1).sum() + (d2 * w2).sum()).backward()
    g1, g2 = t1.grad.cpu().numpy(), t2.grad.cpu().numpy()
    i1n, i2n = i1.cpu().numpy(), i2.cpu().numpy()
    e1, e2 = oracle.chamfer_backward(x1, x2, w1.cpu().numpy(), w2.cpu().numpy(), i1n, i2n)
    f1, f2 = bf.chamfer_grad64(x1, x2, w1.cpu().numpy(), w2.cpu().numpy(), i1n, i2n)
    scale = max(np.abs(f1).max(), np.abs(f2).max())
    # fp32 atomics: order-dependent rounding -> tolerance, not bits (rtol 1e-5 of the fp64 formula)
    assert np.abs(g1 - f1).max() <= 1e-5 * scale and np.abs(g2 - f2).max() <= 1e-5 * scale
    assert np.abs(g1 - e1).max() <= 1e-5 * scale and np.abs(g2 - e2).max() <= 1e-5 * scale


def test_backward_equals_torch_autograd(cuda):
    """Gradient of sum(dist) through the op == torch autograd of the same formula on fixed idx."""
    from pytorch_points_amd.network.model_loss import nndistance
    x1, x2 = _clouds(2, 500, 400, 3)
    t1 = torch.from_numpy(x1).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(x2).to(cuda).requires_grad_(True)
    d1, d2, i1, i2 = nndistance(t1, t2)
    (d1.mean() + d2.mean()).backward()
    a1 = t1.detach().double().requires_grad_(True)
    a2 = t2.detach().double().requires_grad_(True)
    r1 = ((a1 - torch.gather(a2, 1, i1.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    r2 = ((a2 - torch.gather(a1, 1, i2.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
    (r1.mean() + r2.mean()).backward()
    assert torch.allclose(t1.grad.double(), a1.grad, rtol=1e-5, atol=1e-9)
    assert torch.allclose(t2.grad.double(), a2.grad, rtol=1e-5, atol=1e-9)


def test_backward_with_one_output_unused(cuda):
    """Only dist1 enters the loss: the gradient of dist2 is undefined (None) and counts as zero."""
    from pytorch_points_amd.network.model_loss import nndistance
    x1, x2 = _clouds(2, 700, 600, 3)
    t1 = torch.from_numpy(x1).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(x2).to(cuda).requires_grad_(True)
    d1, d2, i1, i2 = nndistance(t1, t2)
    d1.sum().backward()
    e1, e2 = oracle.chamfer_backward(x1, x2, np.ones((2, 700), np.float32), np.zeros((2, 600), np.float32),
                                     i1.cpu().numpy(), i2.cpu().numpy())
    assert np.allclose(t1.grad.cpu().numpy(), e1, rtol=1e-5, atol=1e-6)
    assert np.allclose(t2.grad.cpu().numpy(), e2, rtol=1e-5, atol=1e-6)


def test_labeled_matches_oracle(cuda, bwd_path):
    from pytorch_points_amd.network.model_loss import labeled_nndistance
    b, n, m = 1, 512, 700
    x1, x2 = _clouds(b, n, m, 3)
    l1 = (S.uniform01(7, (b, n)).reshape(b, n) * 4).astype(np.int64)          # labels 0..3
    l2 = (S.uniform01(8, (b, m)).reshape(b, m) * 3).astype(np.int64)          # label 3 missing
    t1 = torch.from_numpy(x1).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(x2).to(cuda).requires_grad_(True)
    d1, d2, i1, i2 = labeled_nndistance(t1, t2, torch.from_numpy(l1).to(cuda), torch.from_numpy(l2).to(cuda))
    e = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    got = [d1.detach().cpu().numpy(), i1.cpu().numpy(), d2.detach().cpu().numpy(), i2.cpu().numpy()]
    for g, x in zip(got, e):
        assert np.array_equal(g, x)
    assert (got[1] == -1).any() and (got[0][got[1] == -1] == 0).all()
    (d1.sum() + d2.sum()).backward()
    e1, e2 = oracle.chamfer_backward(x1, x2, np.ones((b, n), np.float32), np.ones((b, m), np.float32), e[1], e[3])
    assert np.allclose(t1.grad.cpu().numpy(), e1, rtol=1e-5, atol=1e-6)
    assert np.allclose(t2.grad.cpu().numpy(), e2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("b,n,m,nl1,nl2", [(2, 1000, 777, 4, 3), (1, 300, 2500, 2, 2), (1, 64, 64, 70, 70), (3, 9, 7, 1, 2)])
def test_labeled_both_kernels(cuda, variant, b, n, m, nl1, nl2):
    """tiled scan with the label filter and the one-lane-per-query kernel == oracle (idx -1 / dist 0
    for queries whose label does not occur on the other side; lowest index on ties)."""
    from pytorch_points_amd import _lib
    from pytorch_points_amd.network.model_loss import labeled_nndistance
    x1, x2 = _clouds(b, n, m, 3, dup=True)
    l1 = (S.uniform01(70, (b, n)).reshape(b, n) * nl1).astype(np.int64)
    l2 = (S.uniform01(71, (b, m)).reshape(b, m) * nl2).astype(np.int64)
    setter = _lib.lib().pp_debug_set_labeled_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(variant)
    try:
        d1, d2, i1, i2 = labeled_nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda),
                                            torch.from_numpy(l1).to(cuda), torch.from_numpy(l2).to(cuda))
    finally:
        setter(0)
    e = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    for g, x in zip([d1, i1, d2, i2], e):
        assert np.array_equal(g.cpu().numpy(), x)


def test_rejects_bad_inputs(cuda):
    from pytorch_points_amd._ext import losses
    x = torch.zeros(1, 4, 3, device=cuda)
    d = torch.zeros(1, 4, device=cuda)
    i = torch.zeros(1, 4, dtype=torch.int32, device=cuda)
    with pytest.raises(RuntimeError):
        losses.nmdistance_forward(x.cpu(), x, d, d, i, i)
    with pytest.raises(RuntimeError):
        losses.nmdistance_forward(x.double(), x.double(), d, d, i, i)       # mixed double / float arguments
    with pytest.raises(TypeError, match="float16"):   # bfloat16 is not in the reference's dispatch
        losses.nmdistance_forward(x.bfloat16(), x.bfloat16(), d.bfloat16(), d.bfloat16(), i, i)
    with pytest.raises(RuntimeError):
        losses.nmdistance_forward(x.half(), x.half(), d, d, i, i)            # mixed half / float arguments
    with pytest.raises(RuntimeError):
        losses.nmdistance_forward(x, x, d, d, i.long(), i)
    with pytest.raises(RuntimeError):
        losses.nmdistance_forward(x, torch.zeros(1, 4, 2, device=cuda), d, d, i, i)
    assert losses.nmdistance_forward(x, x, d, d.clone(), i, i.clone()) == 1


def test_full_size_c2_properties(cuda):
    """BASELINE config 2 (B=32, N=M=16384): full comparison with the oracle on 2 of the 32 batch
    elements, and size-independent properties on all of them."""
    from pytorch_points_amd.network.model_loss import nndistance
    B, N = 32, 16384
    x1 = S.unit_sphere(0, B, N)
    x2 = S.unit_sphere(1, B, N)
    t1 = torch.from_numpy(x1).to(cuda)
    t2 = torch.from_numpy(x2).to(cuda)
    d1, d2, i1, i2 = nndistance(t1, t2)
    sel = [0, 31]
    e = oracle.chamfer_forward(x1[sel], x2[sel])
    assert np.array_equal(d1[sel].cpu().numpy(), e[0]) and np.array_equal(i1[sel].cpu().numpy(), e[1])
    assert np.array_equal(d2[sel].cpu().numpy(), e[2]) and np.array_equal(i2[sel].cpu().numpy(), e[3])
    # (1) the returned distance is the distance to the returned index
    nb = torch.gather(t2, 1, i1.long()[..., None].expand(-1, -1, 3))
    assert torch.allclose(((t1 - nb) ** 2).sum(-1), d1, rtol=1e-5, atol=1e-7)
    # (2) batch elements are independent: a shard gives the same bits as the whole
    s1, s2, j1, j2 = nndistance(t1[8:12].contiguous(), t2[8:12].contiguous())
    assert torch.equal(s1, d1[8:12]) and torch.equal(j1, i1[8:12]) and torch.equal(s2, d2[8:12]) and torch.equal(j2, i2[8:12])
    # (3) swapping the clouds swaps the outputs
    r1, r2, k1, k2 = nndistance(t2, t1)
    assert torch.equal(r1, d2) and torch.equal(k1, i2) and torch.equal(r2, d1) and torch.equal(k2, i1)
    # (4) permuting the reference cloud permutes indices and keeps distances (no ties in this data)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0)).to(cuda)
    p1, _, q1, _ = nndistance(t1[:4].contiguous(), t2[:4, perm].contiguous())
    assert torch.equal(p1, d1[:4]) and torch.equal(perm[q1.long()], i1[:4].long())
    # (5) deterministic
    u1, u2, v1, v2 = nndistance(t1, t2)
    assert torch.equal(u1, d1) and torch.equal(v1, i1) and torch.equal(u2, d2) and torch.equal(v2, i2)


def test_graph_replay_equals_eager(cuda):
    """hipGraph replay of a fixed-shape step (graphs.py): same outputs as the eager operator, and
    inputs rewritten in place between replays are honoured."""
    from pytorch_points_amd.graphs import GraphedChamferStep
    from pytorch_points_amd.network.model_loss import nndistance
    b, n, m = 8, 8192, 4096          # large enough for the grid search to be the automatic choice
    gs = GraphedChamferStep(b, n, m, cuda)
    for seed in (0, 5):
        x1 = torch.from_numpy(S.unit_sphere(seed, b, n)).to(cuda)
        x2 = torch.from_numpy(S.unit_sphere(seed + 1, b, m)).to(cuda)
        g1 = torch.from_numpy(S.normal(seed + 2, (b, n))).to(cuda)
        g2 = torch.from_numpy(S.normal(seed + 3, (b, m))).to(cuda)
        with torch.no_grad():
            gs.xyz1.copy_(x1); gs.xyz2.copy_(x2); gs.grad_dist1.copy_(g1); gs.grad_dist2.copy_(g2)
        d1, d2, i1, i2, gx1, gx2 = gs.replay()
        t1 = x1.clone().requires_grad_(True)
        t2 = x2.clone().requires_grad_(True)
        e1, e2, j1, j2 = nndistance(t1, t2)
        torch.autograd.backward([e1, e2], [g1, g2])
        assert torch.equal(d1, e1) and torch.equal(d2, e2) and torch.equal(i1, j1) and torch.equal(i2, j2)
        assert torch.allclose(gx1, t1.grad, rtol=1e-6, atol=1e-7) and torch.allclose(gx2, t2.grad, rtol=1e-6, atol=1e-7)


def test_native_autograd_nodes_equal_the_python_functions(cuda):
    """nndistance / labeled_nndistance are the C++ autograd nodes of csrc/torch_bridge.cpp; the Python classes
    NmDistanceFunction / LabeledNmdistanceFunction (the reference's class names) are the same operators over the
    same C ABI: outputs and non-differentiable indices agree bit for bit, gradients to 1e-5 (the default backward
    sums in no fixed order), on a brute-force shape and on a grid-search shape."""
    from pytorch_points_amd.network import model_loss as ml
    for b, n, m in ((2, 300, 500), (2, 8192, 8192)):
        x1n, x2n = S.unit_sphere(70, b, n), S.unit_sphere(71, b, m)
        l1 = torch.from_numpy((S.uniform01(72, (b, n)) * 3).astype(np.int64).reshape(b, n)).to(cuda)
        l2 = torch.from_numpy((S.uniform01(73, (b, m)) * 3).astype(np.int64).reshape(b, m)).to(cuda)
        for labeled in (False, True):
            res = []
            for fn in ((ml.labeled_nndistance, ml.LabeledNmdistanceFunction.apply) if labeled
                       else (ml.nndistance, ml.NmDistanceFunction.apply)):
                x1 = torch.from_numpy(x1n).to(cuda).requires_grad_(True)
                x2 = torch.from_numpy(x2n).to(cuda).requires_grad_(True)
                out = fn(x1, x2, l1, l2) if labeled else fn(x1, x2)
                d1, d2, i1, i2 = out
                assert d1.requires_grad and not i1.requires_grad and i1.dtype == torch.int32
                (d1.sum() * 0.5 + (d2 * d2).sum()).backward()
                res.append((d1.detach(), d2.detach(), i1, i2, x1.grad, x2.grad))
            for k, (a, c) in enumerate(zip(*res)):
                if k < 4:
                    assert torch.equal(a, c)
                else:   # gradients: same terms, summation order of the default backward is not fixed (1e-5)
                    assert torch.allclose(a, c, rtol=1e-5, atol=1e-6)   # (sums of terms of either sign: absolute floor)
    # a missing upstream gradient (only dist1 used) counts as zero in both
    x1 = torch.from_numpy(S.unit_sphere(74, 1, 64)).to(cuda).requires_grad_(True)
    x2 = torch.from_numpy(S.unit_sphere(75, 1, 80)).to(cuda).requires_grad_(True)
    ml.nndistance(x1, x2)[0].sum().backward()
    ga, gb = x1.grad.clone(), x2.grad.clone()
    x1.grad = x2.grad = None
    ml.NmDistanceFunction.apply(x1, x2)[0].sum().backward()
    assert torch.allclose(ga, x1.grad, rtol=1e-5, atol=1e-9) and torch.allclose(gb, x2.grad, rtol=1e-5, atol=1e-9)
    with pytest.raises(TypeError, match="float16"):
        ml.nndistance(x1.bfloat16(), x2.bfloat16())
    with pytest.raises(TypeError):   # labeled Chamfer serves float32 only
        ml.LabeledNmdistanceFunction.apply(x1.half(), x2.half(), torch.zeros(1, 64, device=cuda), torch.zeros(1, 80, device=cuda))
    with pytest.raises(RuntimeError, match="disagree"):
        ml.nndistance(x1, torch.zeros(2, 4, 3, device=cuda))


@pytest.mark.parametrize("shape", [(2, 1000, 777, 1), (2, 1024, 3000, 2), (1, 2049, 513, 4), (1, 700, 1300, 5),
                                   (2, 999, 1024, 6), (1, 515, 2100, 7), (1, 4096, 300, 8), (1, 600, 900, 9),
                                   (2, 1000, 1000, 12), (1, 257, 1025, 16)])
@pytest.mark.parametrize("variant", [0, 9])
def test_forward_other_point_dimensions_tiled_and_plain(cuda, shape, variant):
    """C != 3 (the reference's kernel is generic in c, nmdistance_cuda.cu:31-35): the tiled kernel (variant 0:
    registers x scalar loads x min3 groups, as for C = 3) and the one-lane-per-query kernel (variant 9), clouds
    with duplicated points (ties -> lowest index) and sizes that are not multiples of the group: bit-exact."""
    from pytorch_points_amd import _lib
    b, n, m, c = shape
    x1, x2 = _clouds(b, n, m, c, dup=True)
    exp = oracle.chamfer_forward(x1, x2)
    setter = _lib.lib().pp_debug_set_nmdistance_variant
    setter.argtypes = [ctypes.c_int]
    setter.restype = None
    setter(variant)
    try:
        got = _run(cuda, x1, x2)
    finally:
        setter(0)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


# ------------------------------------------------------------------ double clouds (reference: scalar_t = double)
@pytest.mark.parametrize("b,n,m,c", [(2, 700, 900, 3), (1, 1, 5, 3), (3, 257, 256, 2), (1, 513, 1030, 5), (2, 64, 64, 1),
                                     (1, 300, 200, 8), (1, 100, 90, 11)])
def test_double_forward_equals_oracle(cuda, b, n, m, c):
    """nndistance on double clouds: the reference dispatches its kernel over the floating types
    (_ext/nmdistance_cuda.cu:125); bit-exact distances and indices against the fp64 restatement."""
    from pytorch_points_amd._ext import losses
    x1 = S.normal(800 + n, (b, n, c)).astype(np.float64) + S.normal(801 + n, (b, n, c)).astype(np.float64) * 1e-9
    x2 = S.normal(802 + m, (b, m, c)).astype(np.float64) + S.normal(803 + m, (b, m, c)).astype(np.float64) * 1e-9
    t1, t2 = torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda)
    d1 = torch.empty(b, n, dtype=torch.float64, device=cuda)
    d2 = torch.empty(b, m, dtype=torch.float64, device=cuda)
    i1 = torch.empty(b, n, dtype=torch.int32, device=cuda)
    i2 = torch.empty(b, m, dtype=torch.int32, device=cuda)
    assert losses.nmdistance_forward(t1, t2, d1, d2, i1, i2) == 1
    e = oracle.chamfer_forward_f64(x1, x2)
    for g, x in zip([d1, i1, d2, i2], e):
        assert np.array_equal(g.cpu().numpy(), x)


def test_double_ties_take_the_lowest_index(cuda):
    from pytorch_points_amd.network.model_loss import nndistance
    x2 = np.zeros((1, 1200, 3), np.float64)
    x2[0, :, 0] = np.arange(1200) % 7          # many exact duplicates, also across the reference's 512-chunks
    x1 = np.zeros((1, 9, 3), np.float64)
    x1[0, :, 0] = np.arange(9) - 1.0
    d1, d2, i1, i2 = nndistance(torch.from_numpy(x1).to(cuda), torch.from_numpy(x2).to(cuda))
    e = oracle.chamfer_forward_f64(x1, x2)
    assert d1.dtype is torch.float64 and i1.dtype is torch.int32
    for g, x in zip([d1, i1, d2, i2], e):
        assert np.array_equal(g.cpu().numpy(), x)
    assert i1[0, 1].item() == 0 and i1[0, 2].item() == 1      # first of the equal candidates


def test_double_backward_and_autograd(cuda):
    """gradients in double: C ABI against the oracle (sums of at most a few terms: 1e-12), and the autograd node
    against torch's own double arithmetic"""
    from pytorch_points_amd._ext import losses
    from pytorch_points_amd.network import model_loss as ml
    b, n, m, c = 2, 600, 450, 3
    x1 = S.normal(810, (b, n, c)).astype(np.float64)
    x2 = S.normal(811, (b, m, c)).astype(np.float64)
    g1 = S.normal(812, (b, n)).astype(np.float64)
    g2 = S.normal(813, (b, m)).astype(np.float64)
    _, i1, _, i2 = oracle.chamfer_forward_f64(x1, x2)
    e1, e2 = oracle.chamfer_backward_f64(x1, x2, g1, g2, i1, i2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    o1 = torch.full((b, n, c), 7.0, dtype=torch.float64, device=cuda)     # overwritten, not accumulated into
    o2 = torch.full((b, m, c), 7.0, dtype=torch.float64, device=cuda)
    assert losses.nmdistance_backward(T(x1), T(x2), o1, o2, T(g1), T(g2), T(i1), T(i2)) == 1
    assert np.allclose(o1.cpu().numpy(), e1, rtol=1e-12, atol=1e-13)
    assert np.allclose(o2.cpu().numpy(), e2, rtol=1e-12, atol=1e-13)
    t1, t2 = T(x1).requires_grad_(True), T(x2).requires_grad_(True)
    d1, d2, j1, j2 = ml.nndistance(t1, t2)
    (d1.sum() + 2 * d2.sum()).backward()
    a1, a2 = T(x1).requires_grad_(True), T(x2).requires_grad_(True)
    r1 = ((a1 - torch.gather(a2, 1, j1.long()[..., None].expand(-1, -1, c))) ** 2).sum(-1)
    r2 = ((a2 - torch.gather(a1, 1, j2.long()[..., None].expand(-1, -1, c))) ** 2).sum(-1)
    (r1.sum() + 2 * r2.sum()).backward()
    assert torch.allclose(d1, r1.detach(), rtol=1e-12, atol=1e-14)
    assert torch.allclose(t1.grad, a1.grad, rtol=1e-11, atol=1e-12) and torch.allclose(t2.grad, a2.grad, rtol=1e-11, atol=1e-12)
    # empty clouds
    z = torch.zeros(1, 0, 3, dtype=torch.float64, device=cuda)
    d1, d2, j1, j2 = ml.nndistance(z, T(x2[:1]))
    assert d1.shape == (1, 0) and (d2 == 0).all() and (j2 == 0).all()


@pytest.mark.parametrize("shape", [(2, 300, 500, 3), (1, 1000, 777, 3), (2, 64, 64, 2), (1, 513, 511, 5), (1, 257, 40, 1),
                                   (1, 130, 900, 11)])
def test_half_clouds_match_the_half_oracle(cuda, shape):
    """scalar_t = at::Half, the third type of the reference's dispatch (nmdistance_cuda.cu:125,210): every operation
    rounded to half separately (c10::Half's operators), comparisons in half, first minimum in index order -- with 11
    bits of significand EXACT TIES are everywhere, so the tie rule is what this checks.  Indices and distances bit for
    bit against oracle.chamfer_forward_f16 (numpy float16 arithmetic); gradients: the own-row terms bit for bit where a
    row receives nothing else, every row within half's rounding of the exact sum of the half-rounded terms (the
    reference adds them with half atomics in no fixed order)."""
    from pytorch_points_amd.network import model_loss as ml
    b, n, m, c = shape
    x1 = (S.unit_sphere(90 + n, b, n, c) * 0.75).astype(np.float16)
    x2 = (S.unit_sphere(91 + m, b, m, c) * 0.75).astype(np.float16)
    x2[:, : min(m, 20)] = x2[:, m - min(m, 20):]                       # exact duplicates: lowest index must win
    e_d1, e_i1, e_d2, e_i2 = oracle.chamfer_forward_f16(x1, x2)
    t1 = torch.from_numpy(x1).to(cuda).requires_grad_(True)
    t2 = torch.from_numpy(x2).to(cuda).requires_grad_(True)
    d1, d2, i1, i2 = ml.nndistance(t1, t2)
    assert d1.dtype == torch.float16 and i1.dtype == torch.int32 and d1.requires_grad and not i1.requires_grad
    assert np.array_equal(i1.cpu().numpy(), e_i1) and np.array_equal(i2.cpu().numpy(), e_i2)
    assert np.array_equal(d1.detach().cpu().numpy().view(np.uint16), e_d1.view(np.uint16))
    assert np.array_equal(d2.detach().cpu().numpy().view(np.uint16), e_d2.view(np.uint16))
    g1 = (np.abs(S.normal(92, (b, n))) * 0.5 + 0.25).astype(np.float16)
    g2 = (np.abs(S.normal(93, (b, m))) * 0.5 + 0.25).astype(np.float16)
    torch.autograd.backward([d1, d2], [torch.from_numpy(g1).to(cuda), torch.from_numpy(g2).to(cuda)])
    own1, own2, s1, s2 = oracle.chamfer_backward_f16_terms(x1, x2, g1, g2, e_i1, e_i2)
    gx1, gx2 = t1.grad.cpu().numpy(), t2.grad.cpu().numpy()
    assert gx1.dtype == np.float16
    for gx, own, ssum, idx_other, nn in ((gx1, own1, s1, e_i2, n), (gx2, own2, s2, e_i1, m)):
        for k in range(b):
            hit = np.zeros(nn, bool)
            hit[idx_other[k]] = True                                   # rows that also receive scattered terms
            assert np.array_equal(gx[k][~hit].view(np.uint16), own[k][~hit].view(np.uint16))
            cnt = np.bincount(idx_other[k], minlength=nn)[:, None] + 1   # additions per row, each rounded to half
            tol = cnt * 2.0 ** -10 * np.maximum(np.abs(ssum[k]), np.abs(own[k].astype(np.float64))) + cnt * 1e-3
            assert np.all(np.abs(gx[k].astype(np.float64) - ssum[k]) <= tol)
    # the extension-module entry points take half tensors as well (the pybind functions' dispatch)
    from pytorch_points_amd._ext import losses
    o = (torch.empty(b, n, dtype=torch.float16, device=cuda), torch.empty(b, m, dtype=torch.float16, device=cuda),
         torch.empty(b, n, dtype=torch.int32, device=cuda), torch.empty(b, m, dtype=torch.int32, device=cuda))
    assert losses.nmdistance_forward(t1.detach(), t2.detach(), *o) == 1
    assert torch.equal(o[0], d1.detach()) and torch.equal(o[2], i1)


def test_half_empty_clouds(cuda):
    from pytorch_points_amd.network import model_loss as ml
    x1 = torch.zeros(2, 0, 3, dtype=torch.float16, device=cuda)
    x2 = torch.rand(2, 5, 3, device=cuda).half()
    d1, d2, i1, i2 = ml.nndistance(x1, x2)
    assert d1.shape == (2, 0) and torch.count_nonzero(d2) == 0 and torch.count_nonzero(i2) == 0
