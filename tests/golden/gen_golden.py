#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs and expected outputs for the hot-path ops.

PROVENANCE (read this): the reference has no tests or fixtures for this path and cannot be built
or imported in this image, so these vectors do NOT come from the reference.  They are outputs of
the CPU oracle (oracle/pp_oracle.c, a line-by-line restatement of the reference kernels), and each
one was accepted only after the independent fp64 brute force in oracle/bruteforce.py found no
disagreement that is not a proven near-tie.  They pin the oracle and the HIP kernels against
regressions and against each other; they do not pin either to the reference ("parity unpinned").

Run from the repo root:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from oracle import bruteforce as bf  # noqa: E402
from pytorch_points_amd import synthetic as S  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, {k: v.shape for k, v in arrays.items()})


def chamfer_case(name, b, n, m, c, dup=False, seed=0):
    x1 = S.unit_sphere(100 + seed, b, n, c)
    x2 = S.unit_sphere(200 + seed, b, m, c)
    if dup:
        x2[:, m // 2:] = x2[:, : m - m // 2]
        x1[:, : min(n, m) // 3] = x2[:, : min(n, m) // 3]
    d1, i1, d2, i2 = oracle.chamfer_forward(x1, x2, structural=True)
    r1, r2 = bf.check_nn(x1, x2, d1, i1), bf.check_nn(x2, x1, d2, i2)
    assert r1["bad_idx"] == 0 and r2["bad_idx"] == 0 and r1["exact_tie_wrong"] == 0 and r2["exact_tie_wrong"] == 0, (r1, r2)
    assert r1["max_rel_err"] < 1e-5 and r2["max_rel_err"] < 1e-5
    gd1 = S.normal(300 + seed, (b, n))
    gd2 = S.normal(400 + seed, (b, m))
    g1, g2 = oracle.chamfer_backward(x1, x2, gd1, gd2, i1, i2)
    f1, f2 = bf.chamfer_grad64(x1, x2, gd1, gd2, i1, i2)
    assert np.allclose(g1, f1, rtol=1e-5, atol=1e-6) and np.allclose(g2, f2, rtol=1e-5, atol=1e-6)
    save(name, xyz1=x1, xyz2=x2, dist1=d1, idx1=i1, dist2=d2, idx2=i2, graddist1=gd1, graddist2=gd2,
         gradxyz1=g1, gradxyz2=g2)


def main():
    chamfer_case("chamfer_c1_b2_n1024_m1024_c3", 2, 1024, 1024, 3)           # BASELINE configs[0]
    chamfer_case("chamfer_b1_n1000_m777_c3", 1, 1000, 777, 3, seed=1)
    chamfer_case("chamfer_b2_n300_m1500_c3_dup", 2, 300, 1500, 3, dup=True, seed=2)
    chamfer_case("chamfer_b1_n64_m64_c2", 1, 64, 64, 2, seed=3)
    chamfer_case("chamfer_b1_n513_m511_c5", 1, 513, 511, 5, seed=4)

    # labeled Chamfer: 4 labels on side 1, label 3 missing on side 2
    b, n, m = 1, 512, 700
    x1, x2 = S.unit_sphere(110, b, n), S.unit_sphere(210, b, m)
    l1 = (S.uniform01(111, (b, n)).reshape(b, n) * 4).astype(np.int32)
    l2 = (S.uniform01(211, (b, m)).reshape(b, m) * 3).astype(np.int32)
    d1, i1, d2, i2 = oracle.labeled_chamfer_forward(x1, x2, l1, l2)
    D = bf.sqdist64(x1, x2)
    D[l1[:, :, None] != l2[:, None, :]] = np.inf
    has = np.isfinite(D).any(-1)
    assert ((i1 >= 0) == has).all() and (D.argmin(-1)[has] == i1[has]).all() and (d1[~has] == 0).all()
    save("labeled_b1_n512_m700", xyz1=x1, xyz2=x2, label1=l1, label2=l2, dist1=d1, idx1=i1, dist2=d2, idx2=i2)

    # furthest point sampling
    for name, b, n, m, seed in [("fps_b2_n2048_m256", 2, 2048, 256, 0), ("fps_b1_n300_m64_seed7", 1, 300, 64, 7),
                                ("fps_b1_n5000_m128", 1, 5000, 128, 0)]:
        x = S.unit_sphere(120 + n, b, n)
        idx, temp = oracle.furthest_sampling(x, m, seed)
        r = bf.check_fps(x, idx, seed)
        assert r["bad"] == 0, r
        save(name, xyz=x, idx=idx, temp=temp, seed=np.int32(seed))

    # ball query (centres = FPS-256 of the cloud) and three_nn
    x = S.unit_sphere(130, 2, 2048)
    fidx, _ = oracle.furthest_sampling(x, 256, 0)
    centres = np.ascontiguousarray(np.take_along_axis(x, fidx[..., None].astype(np.int64), 1))
    arrays = dict(xyz=x, new_xyz=centres)
    for r in (0.05, 0.2, 0.5):
        for ns in (16, 64):
            idx = oracle.ball_query(centres, x, r, ns)
            chk = bf.check_ball_query(centres, x, r, ns, idx)
            assert chk["bad_rows"] == 0, chk
            arrays["idx_r%g_ns%d" % (r, ns)] = idx
    save("ball_query_b2_n2048_m256", **arrays)

    for name, b, n, m in [("three_nn_b2_n2048_m256", 2, 2048, 256), ("three_nn_b1_n10_m2", 1, 10, 2)]:
        u, k = S.unit_sphere(140 + n, b, n), S.unit_sphere(150 + m, b, m)
        d2, idx = oracle.three_nn(u, k)
        chk = bf.check_three_nn(u, k, d2, idx)
        assert chk["bad"] == 0 and chk["max_rel_err"] < 1e-5, chk
        save(name, unknown=u, known=k, dist2=d2, idx=idx)


def knn_case():
    """K nearest neighbours: oracle.knn accepted by an fp64 evaluation"""
    u, k, K = S.unit_sphere(160, 2, 600), S.unit_sphere(161, 2, 500), 8
    k[:, 250:260] = k[:, :10]            # exact ties: the lower index first
    d2, idx = oracle.knn(u, k, K)
    D = ((u[:, :, None].astype(np.float64) - k[:, None].astype(np.float64)) ** 2).sum(-1)
    ref = np.argsort(D, axis=-1, kind="stable")[..., :K]
    dref = np.take_along_axis(D, ref, -1)
    assert np.allclose(d2, dref, rtol=1e-5, atol=1e-7)
    picked = np.take_along_axis(D, idx.astype(np.int64), -1)
    assert np.allclose(picked, dref, rtol=1e-5, atol=1e-7)       # same neighbours up to fp32 near-ties
    save("knn_b2_n600_m500_k8", p1=u, p2=k, dist2=d2, idx=idx, K=np.int32(K))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "knn":   # add the K-NN fixture without rewriting the others
        knn_case()
    else:
        main()
        knn_case()
