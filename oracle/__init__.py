"""CPU oracle for the pytorch_points `_ext` hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference holds no tests/golden vectors for this path and cannot be built or
imported in this image (see oracle/pp_oracle.c header and DESIGN.md).  The C restatement in
``pp_oracle.c`` follows the reference kernels line by line and is cross-checked against the
independent fp64 brute force in ``oracle/bruteforce.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  Nothing in ``pytorch_points_amd`` does.

numpy in, numpy out; every wrapper mirrors one C function of ``pp_oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "pp_oracle.c")
_SO = os.path.join(_HERE, "libpp_oracle.so")

# x86-64-v3 (AVX2 + FMA) rather than -march=native: the .so is built in the dev container and
# travels to the GPU box, whose host CPU may be a different micro-architecture.
# -ffp-contract=off: every fused multiply-add in the canonical arithmetic is an explicit fmaf.
_CFLAGS = ["-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
           "-fPIC", "-shared", "-std=c11", "-Wall"]


def build(force=False):
    """Compile pp_oracle.c -> libpp_oracle.so with gcc (no-op when up to date)."""
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(_SRC)):
        return _SO
    cmd = ["gcc", *_CFLAGS, _SRC, "-o", _SO, "-lm"]
    subprocess.run(cmd, check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or (os.path.exists(_SRC) and
                                       os.path.getmtime(_SO) < os.path.getmtime(_SRC)):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_opt_n_threads.restype = ctypes.c_int
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().oracle_num_threads())


def opt_n_threads(work_size):
    return int(lib().oracle_opt_n_threads(ctypes.c_int(int(work_size))))


def chamfer_forward(xyz1, xyz2, structural=False):
    """-> dist1 (B,N) f32, idx1 (B,N) i32, dist2 (B,M) f32, idx2 (B,M) i32"""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, c = xyz1.shape
    _, m, c2 = xyz2.shape
    assert c == c2 and xyz2.shape[0] == b
    # zeros: what the reference's Python wrapper allocates (network/model_loss.py:412-416); they
    # survive only when n == 0 or m == 0.
    d1 = np.zeros((b, n), np.float32)
    d2 = np.zeros((b, m), np.float32)
    i1 = np.zeros((b, n), np.int32)
    i2 = np.zeros((b, m), np.int32)
    fn = lib().oracle_chamfer_forward_structural if structural else lib().oracle_chamfer_forward
    fn(p1, p2, _p(d1), _p(i1), _p(d2), _p(i2), b, n, m, c)
    return d1, i1, d2, i2


def chamfer_forward_f64(xyz1, xyz2):
    """the reference's kernel instantiated for double (nmdistance_cuda.cu:125): -> dist1 f64, idx1 i32, dist2, idx2"""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float64)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float64)
    b, n, c = xyz1.shape
    _, m, c2 = xyz2.shape
    assert c == c2 and xyz2.shape[0] == b
    d1 = np.zeros((b, n), np.float64)
    d2 = np.zeros((b, m), np.float64)
    i1 = np.zeros((b, n), np.int32)
    i2 = np.zeros((b, m), np.int32)
    lib().oracle_chamfer_forward_f64(_p(xyz1), _p(xyz2), _p(d1), _p(i1), _p(d2), _p(i2), b, n, m, c)
    return d1, i1, d2, i2


def chamfer_backward_f64(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """-> gradxyz1 (B,N,C), gradxyz2 (B,M,C), double (nmdistance_cuda.cu:210)"""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float64)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float64)
    graddist1 = np.ascontiguousarray(graddist1, dtype=np.float64)
    graddist2 = np.ascontiguousarray(graddist2, dtype=np.float64)
    idx1, q1 = _i(idx1)
    idx2, q2 = _i(idx2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    gx1 = np.zeros_like(xyz1)
    gx2 = np.zeros_like(xyz2)
    lib().oracle_chamfer_backward_f64(_p(xyz1), _p(xyz2), _p(graddist1), _p(graddist2), q1, q2, _p(gx1), _p(gx2), b, n, m, c)
    return gx1, gx2


def chamfer_forward_f16(xyz1, xyz2):
    """the reference's kernel instantiated for at::Half (nmdistance_cuda.cu:7-49,125): every operator of c10::Half
    converts to float, computes, and rounds the result back to half (c10/util/Half-inl.h) -- tmp = buf - xyz; d += tmp *
    tmp is three separately rounded half operations per coordinate, which is what numpy's float16 arithmetic does
    (computed in float32, rounded to float16: exact products, innocuous double rounding of sums); first minimum in
    index order.  -> dist1 f16, idx1 i32, dist2 f16, idx2 i32.  Plain numpy, small sizes."""
    xyz1 = np.ascontiguousarray(xyz1, dtype=np.float16)
    xyz2 = np.ascontiguousarray(xyz2, dtype=np.float16)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]

    def one_way(q, r):          # q (nq, c) queries, r (nr, c) references
        d = np.zeros((q.shape[0], r.shape[0]), np.float16)
        for e in range(c):
            t = (r[None, :, e] - q[:, None, e]).astype(np.float16)       # buf - xyz (:33)
            p = (t * t).astype(np.float16)
            d = p if e == 0 else (d + p).astype(np.float16)            # 0 + p = p exactly
        best = d.min(axis=1) if r.shape[0] else np.zeros(q.shape[0], np.float16)
        idx = d.argmin(axis=1).astype(np.int32) if r.shape[0] else np.zeros(q.shape[0], np.int32)   # first minimum
        return best.astype(np.float16), idx

    d1 = np.zeros((b, n), np.float16); d2 = np.zeros((b, m), np.float16)
    i1 = np.zeros((b, n), np.int32); i2 = np.zeros((b, m), np.int32)
    for k in range(b):
        if n and m:
            d1[k], i1[k] = one_way(xyz1[k], xyz2[k])
            d2[k], i2[k] = one_way(xyz2[k], xyz1[k])
    return d1, i1, d2, i2


def chamfer_backward_f16_terms(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """the half instantiation of the backward kernel (:168-185): g = graddist * 2; v = g * (xa - xb), every operation
    rounded to half; -> (own1 (B,N,C), own2 (B,M,C)) the own-row terms as half, and float64 sums of all terms per row
    (gx1, gx2): the reference adds the scattered terms with half atomics in arrival order, so only a tolerance against
    the exact sum of the half-rounded TERMS is defined."""
    x1 = np.ascontiguousarray(xyz1, dtype=np.float16); x2 = np.ascontiguousarray(xyz2, dtype=np.float16)
    g1 = np.ascontiguousarray(graddist1, dtype=np.float16); g2 = np.ascontiguousarray(graddist2, dtype=np.float16)
    b, n, c = x1.shape
    m = x2.shape[1]
    two = np.float16(2)
    own1 = np.zeros((b, n, c), np.float16); own2 = np.zeros((b, m, c), np.float16)
    gx1 = np.zeros((b, n, c), np.float64); gx2 = np.zeros((b, m, c), np.float64)
    for k in range(b):
        ga = (g1[k] * two).astype(np.float16)
        v = (ga[:, None] * (x1[k] - x2[k][idx1[k]]).astype(np.float16)).astype(np.float16)
        own1[k] = v
        gx1[k] += v.astype(np.float64)
        np.subtract.at(gx2[k], idx1[k], v.astype(np.float64))
        gb = (g2[k] * two).astype(np.float16)
        w = (gb[:, None] * (x2[k] - x1[k][idx2[k]]).astype(np.float16)).astype(np.float16)
        own2[k] = w
        gx2[k] += w.astype(np.float64)
        np.subtract.at(gx1[k], idx2[k], w.astype(np.float64))
    return own1, own2, gx1, gx2


def labeled_chamfer_forward(xyz1, xyz2, label1, label2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    label1, l1 = _f(label1)  # labels are cast to the xyz dtype (network/model_loss.py:452-453)
    label2, l2 = _f(label2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.zeros((b, n), np.float32)
    d2 = np.zeros((b, m), np.float32)
    i1 = np.zeros((b, n), np.int32)
    i2 = np.zeros((b, m), np.int32)
    lib().oracle_labeled_chamfer_forward(p1, p2, l1, l2, _p(d1), _p(i1), _p(d2), _p(i2), b, n, m, c)
    return d1, i1, d2, i2


def chamfer_backward(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """-> gradxyz1 (B,N,C), gradxyz2 (B,M,C)"""
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    graddist1, g1 = _f(graddist1)
    graddist2, g2 = _f(graddist2)
    idx1, q1 = _i(idx1)
    idx2, q2 = _i(idx2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    gx1 = np.zeros_like(xyz1)
    gx2 = np.zeros_like(xyz2)
    lib().oracle_chamfer_backward(p1, p2, g1, g2, q1, q2, _p(gx1), _p(gx2), b, n, m, c)
    return gx1, gx2


def furthest_sampling(xyz, npoint, seed_idx=0, temp=None):
    """-> idx (B,npoint) i32, temp (B,N) f32 after the call"""
    xyz, px = _f(xyz)
    b, n, three = xyz.shape
    assert three == 3
    if temp is None:
        temp = np.full((b, n), 1e10, np.float32)  # network/geo_operations.py:33
    temp = np.array(temp, dtype=np.float32, copy=True, order="C")
    idx = np.zeros((b, npoint), np.int32)
    lib().oracle_furthest_sampling(px, _p(temp), _p(idx), b, n, int(npoint), int(seed_idx))
    return idx, temp


def gather_forward(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().oracle_gather_forward(pp, pi, _p(out), b, c, n, m)
    return out


def gather_backward(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = grad_out.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().oracle_gather_backward(pg, pi, _p(gp), b, c, int(n), m)
    return gp


def ball_query(new_xyz, xyz, radius, nsample):
    """argument order of the native function (sampling.cpp:85): (new_xyz, xyz, radius, nsample)"""
    new_xyz, pn = _f(new_xyz)
    xyz, px = _f(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().oracle_ball_query(pn, px, _p(idx), b, n, m, ctypes.c_float(radius), int(nsample))
    return idx


def group_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().oracle_group_points(pp, pi, _p(out), b, c, n, npoints, nsample)
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, npoints, nsample = grad_out.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().oracle_group_points_grad(pg, pi, _p(gp), b, c, int(n), npoints, nsample)
    return gp


def three_nn(unknown, known):
    """-> dist2 (B,N,3) f32 (squared, before the wrapper's sqrt), idx (B,N,3) i32"""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().oracle_three_nn(pu, pk, _p(d2), _p(idx), b, n, m)
    return d2, idx


def knn(p1, p2, K, lengths1=None, lengths2=None):
    """K nearest neighbours of p1 (B,N,D) in p2 (B,M,D) -> dist2 (B,N,K) f32 ascending, idx (B,N,K) i32;
    squared distance = sequential fma chain over the D dimensions; ties to the lower index; padding (0, 0)"""
    p1, q1 = _f(p1)
    p2, q2 = _f(p2)
    b, n, dim = p1.shape
    m = p2.shape[1]
    assert p2.shape[2] == dim
    d2 = np.zeros((b, n, K), np.float32)
    idx = np.zeros((b, n, K), np.int32)
    l1 = l2 = None
    a1 = a2 = None
    if lengths1 is not None:
        a1 = np.ascontiguousarray(lengths1, np.int32)
        l1 = a1.ctypes.data_as(ctypes.c_void_p)
    if lengths2 is not None:
        a2 = np.ascontiguousarray(lengths2, np.int32)
        l2 = a2.ctypes.data_as(ctypes.c_void_p)
    lib().oracle_knn_nd(q1, q2, l1, l2, _p(d2), _p(idx), b, n, m, int(dim), int(K))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().oracle_three_interpolate(pp, pi, pw, _p(out), b, c, m, n)
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, n = grad_out.shape
    gp = np.zeros((b, c, m), np.float32)
    lib().oracle_three_interpolate_grad(pg, pi, pw, _p(gp), b, c, n, int(m))
    return gp
