"""Independent fp64 brute-force checkers -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

These do not restate the reference's loops; they state the mathematical result (nearest neighbour,
farthest point, in-radius set, three nearest) in float64 numpy and classify any disagreement with
an fp32 implementation as a *near-tie* (the competing fp64 values differ by no more than a few fp32
ulps of the value) or a *bug*.  They are what keeps pp_oracle.c honest in the absence of golden
vectors from the reference.
"""
import numpy as np

# a disagreement is a near-tie if the fp64 quantities differ by < TIE_ULPS fp32 ulps of their size
TIE_ULPS = 8.0


def _ulp32(x):
    x = np.maximum(np.abs(np.asarray(x, np.float64)), np.finfo(np.float32).tiny)
    return np.spacing(x.astype(np.float32)).astype(np.float64)


def sqdist64(a, b):
    """(B,N,C),(B,M,C) fp32 -> (B,N,M) fp64 squared distances."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)


def check_nn(query, ref, dist, idx, rtol=1e-5, atol=1e-12):
    """Nearest neighbour of each query in ref.  Returns dict(bad_idx, near_ties, max_rel_err)."""
    D = sqdist64(query, ref)
    true_i = D.argmin(-1)
    true_d = np.take_along_axis(D, true_i[..., None], -1)[..., 0]
    got_d64 = np.take_along_axis(D, np.asarray(idx, np.int64)[..., None], -1)[..., 0]
    differ = true_i != idx
    tie = differ & (np.abs(got_d64 - true_d) <= TIE_ULPS * _ulp32(true_d))
    # exact fp64 ties must resolve to the lowest index
    exact_tie_wrong = differ & (got_d64 == true_d) & (np.asarray(idx) > true_i)
    rel = np.abs(np.asarray(dist, np.float64) - got_d64) / np.maximum(got_d64, atol / rtol)
    return dict(bad_idx=int((differ & ~tie).sum()), near_ties=int(tie.sum()),
                exact_tie_wrong=int(exact_tie_wrong.sum()), max_rel_err=float(rel.max(initial=0.0)))


def chamfer_grad64(xyz1, xyz2, gd1, gd2, idx1, idx2):
    """fp64 evaluation of the reference's backward formula for GIVEN indices."""
    x1 = np.asarray(xyz1, np.float64)
    x2 = np.asarray(xyz2, np.float64)
    g1 = np.zeros_like(x1)
    g2 = np.zeros_like(x2)
    B = x1.shape[0]
    for b in range(B):
        v = idx1[b] >= 0
        t = 2.0 * np.asarray(gd1[b], np.float64)[v, None] * (x1[b][v] - x2[b][idx1[b][v]])
        g1[b][v] += t
        np.add.at(g2[b], idx1[b][v], -t)
        v = idx2[b] >= 0
        t = 2.0 * np.asarray(gd2[b], np.float64)[v, None] * (x2[b][v] - x1[b][idx2[b][v]])
        g2[b][v] += t
        np.add.at(g1[b], idx2[b][v], -t)
    return g1, g2


def check_fps(xyz, idx, seed_idx=0):
    """Every pick must be (within a near-tie) the point farthest from the picks before it.
    Follows the *given* picks, so one near-tie does not cascade.  Returns dict(bad, near_ties)."""
    x = np.asarray(xyz, np.float64)
    B, N, _ = x.shape
    bad = ties = 0
    for b in range(B):
        assert idx[b, 0] == seed_idx
        mind = np.full(N, np.inf)
        for j in range(1, idx.shape[1]):
            old = idx[b, j - 1]
            mind = np.minimum(mind, ((x[b] - x[b, old]) ** 2).sum(-1))
            best = mind.max()
            got = mind[idx[b, j]]
            if idx[b, j] != mind.argmax():
                if abs(best - got) <= TIE_ULPS * _ulp32(best):
                    ties += 1
                else:
                    bad += 1
    return dict(bad=bad, near_ties=ties)


def check_ball_query(new_xyz, xyz, radius, nsample, idx):
    """Row j must list, ascending, the first nsample points strictly inside the ball, padded with
    the first hit (all zeros if none).  Points within a near-tie of the surface may go either way.
    Returns dict(bad_rows, borderline_rows)."""
    D = sqdist64(new_xyz, xyz)
    r2 = float(np.float32(radius) * np.float32(radius))
    band = TIE_ULPS * _ulp32(r2)
    B, M, N = D.shape
    bad = borderline = 0
    for b in range(B):
        for j in range(M):
            d = D[b, j]
            sure = np.nonzero(d < r2 - band)[0]
            maybe = np.nonzero(np.abs(d - r2) <= band)[0]
            row = np.asarray(idx[b, j])
            if len(maybe) == 0:
                hits = sure[:nsample]
                exp = np.zeros(nsample, np.int64)
                if len(hits):
                    exp[:] = hits[0]
                    exp[:len(hits)] = hits
                bad += int(not np.array_equal(exp, row))
            else:
                borderline += 1
                # every listed index must be sure-or-maybe, ascending until the pad
                ok = np.isin(row, np.concatenate([sure, maybe, [0]])).all()
                bad += int(not ok)
    return dict(bad_rows=bad, borderline_rows=borderline)


def check_three_nn(unknown, known, dist2, idx, rtol=1e-5):
    """Three nearest, ascending.  Returns dict(bad, near_ties, max_rel_err)."""
    D = sqdist64(unknown, known)
    B, N, M = D.shape
    k = min(3, M)
    order = np.argsort(D, axis=-1, kind="stable")[..., :k]
    true_d = np.take_along_axis(D, order, -1)
    got_i = np.asarray(idx, np.int64)[..., :k]
    got_d = np.take_along_axis(D, got_i, -1)
    differ = got_i != order
    tie = differ & (np.abs(got_d - true_d) <= TIE_ULPS * _ulp32(true_d))
    rel = np.abs(np.asarray(dist2, np.float64)[..., :k] - got_d) / np.maximum(got_d, 1e-7)
    return dict(bad=int((differ & ~tie).sum()), near_ties=int(tie.sum()),
                max_rel_err=float(rel.max(initial=0.0)))
