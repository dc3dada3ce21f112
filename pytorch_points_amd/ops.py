"""K nearest neighbours with the call signature of ``pytorch3d.ops.knn_points`` -- the one function of
pytorch3d the reference uses (network/model_loss.py:120,147,378, geo_operations.py:112,139,
layers.py:52,99,115), so that those modules can run without the un-vendored dependency
(SURVEY.md §8f N4).

    dists, idx, nn = knn_points(p1, p2, K=8, return_nn=True)

dists (N, P1, K): squared distances, ascending; idx (N, P1, K) int64; nn (N, P1, K, D) or None.
Any point dimension D up to 512 and any K up to 128: 3-D clouds with K <= 32 go through the exact grid search
(csrc/knn.hip: knn_grid_kernel), everything else -- the reference's feature-space searches, layers.py:52,99
(DenseEdgeConv: D = channel count, K = k + 1) -- through the brute force with the sequential distance chain
(knn_nd_kernel).  Differences from pytorch3d, all documented: ties go to the lower index (pytorch3d leaves
them unspecified), ``version`` is accepted and ignored, ``return_sorted=False`` still returns sorted neighbours.
Parity with pytorch3d itself is unpinned (it is absent from the reference tree and this image); the
operator is tested against a brute-force restatement of its published contract (oracle.knn).
"""
from collections import namedtuple

import torch

from . import _lib
from ._ext.sampling import _named_workspace

_KNN = namedtuple("KNN", "dists idx knn")


def _knn_forward(p1, p2, lengths1, lengths2, K):
    dev = _lib.require_cuda(("p1", p1), ("p2", p2))
    _lib.require_float(("p1", p1), ("p2", p2))
    if p1.dim() != 3 or p2.dim() != 3 or p1.size(0) != p2.size(0):
        raise ValueError("p1 and p2 must be (N, P1, D) and (N, P2, D)")
    dim = p1.size(2)
    if p2.size(2) != dim:
        raise ValueError("p1 and p2 must have the same point dimension")
    if not 1 <= dim <= 512:
        raise NotImplementedError("knn_points: 1 <= D <= 512")
    if not 1 <= K <= 128:
        raise NotImplementedError("knn_points: 1 <= K <= 128")
    if not (dim == 3 and K <= 32):
        kt = next(v for v in (4, 8, 16, 32, 64, 128) if K <= v)
        if ((dim + 7) // 8 * 8) * 256 + kt * 512 > 160 * 1024:   # staged queries + one K-list in transit (LDS)
            raise NotImplementedError("knn_points: D = %d with K = %d does not fit the LDS of the brute-force kernel" % (dim, K))
    p1 = p1.contiguous()
    p2 = p2.contiguous()
    b, n, _ = p1.shape
    m = p2.size(1)
    l1 = l2 = None
    if lengths1 is not None:
        l1 = lengths1.to(device=dev, dtype=torch.int32).contiguous()
        if l1.numel() != b:
            raise ValueError("lengths1 must have shape (N,)")
    if lengths2 is not None:
        l2 = lengths2.to(device=dev, dtype=torch.int32).contiguous()
        if l2.numel() != b:
            raise ValueError("lengths2 must have shape (N,)")
    dist = torch.empty(b, n, K, dtype=torch.float32, device=dev)
    idx = torch.empty(b, n, K, dtype=torch.int32, device=dev)
    with _lib.on_device(dev) as stream:
        lp1, lp2 = (_lib.ptr(l1) if l1 is not None else None), (_lib.ptr(l2) if l2 is not None else None)
        if dim == 3 and K <= 32:
            nbytes = int(_lib.lib().pp_knn_workspace_bytes(b, n, m, K))
            ws = _named_workspace(dev, "knn", nbytes)
            _lib.check(_lib.lib().pp_knn_ws_f32(
                _lib.ptr(p1), _lib.ptr(p2), lp1, lp2, _lib.ptr(dist), _lib.ptr(idx), b, n, m, K,
                _lib.ptr(ws) if ws is not None else None, nbytes, stream), "knn_points")
        else:
            _lib.check(_lib.lib().pp_knn_nd_f32(
                _lib.ptr(p1), _lib.ptr(p2), lp1, lp2, _lib.ptr(dist), _lib.ptr(idx), b, n, m, dim, K, stream),
                "knn_points")
    return dist, idx


def _valid_mask(idx, lengths1, lengths2, m):
    """(N, P1, K) bool: slots that hold a real neighbour"""
    b, n, k = idx.shape
    mask = torch.ones(b, n, k, dtype=torch.bool, device=idx.device)
    if lengths2 is not None:
        mask &= torch.arange(k, device=idx.device)[None, None, :] < lengths2.to(idx.device).clamp(max=m)[:, None, None]
    elif k > m:
        mask &= torch.arange(k, device=idx.device)[None, None, :] < m
    if lengths1 is not None:
        mask &= torch.arange(n, device=idx.device)[None, :, None] < lengths1.to(idx.device)[:, None, None]
    return mask


class _KnnFunction(torch.autograd.Function):
    """dists is differentiable w.r.t. p1 and p2: d/dp1 = 2 (p1 - p2[idx]), d/dp2[idx] = -that."""

    @staticmethod
    def forward(ctx, p1, p2, lengths1, lengths2, K):
        dist, idx = _knn_forward(p1, p2, lengths1, lengths2, K)
        idx64 = idx.long()
        ctx.save_for_backward(p1, p2, idx64)
        ctx.lengths = (lengths1, lengths2)
        ctx.mark_non_differentiable(idx64)
        return dist, idx64

    @staticmethod
    def backward(ctx, grad_dist, _grad_idx):
        p1, p2, idx = ctx.saved_tensors
        b, n, k = idx.shape
        m = p2.size(1)
        mask = _valid_mask(idx, ctx.lengths[0], ctx.lengths[1], m)
        g = (grad_dist * mask).unsqueeze(-1) * 2.0                                   # (b, n, k, 1)
        dim = p1.size(2)
        nb = torch.gather(p2.unsqueeze(1).expand(-1, n, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, dim))
        diff = g * (p1.unsqueeze(2) - nb)                                            # (b, n, k, D)
        grad_p1 = diff.sum(2)
        grad_p2 = torch.zeros_like(p2)
        grad_p2.scatter_add_(1, idx.reshape(b, n * k, 1).expand(-1, -1, dim), -diff.reshape(b, n * k, dim))
        return grad_p1, grad_p2, None, None, None


def knn_gather(x, idx, lengths=None):
    """pytorch3d.ops.knn_gather: x (N, M, U), idx (N, L, K) -> (N, L, K, U); slots beyond ``lengths`` are 0."""
    n, m, u = x.shape
    _, l, k = idx.shape
    out = torch.gather(x.unsqueeze(1).expand(-1, l, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, u))
    if lengths is not None:
        keep = torch.arange(k, device=x.device)[None, None, :] < lengths.to(x.device)[:, None, None]
        out = out * keep.unsqueeze(-1)
    elif k > m:
        out = out * (torch.arange(k, device=x.device)[None, None, :, None] < m)
    return out


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False, return_sorted=True):
    """pytorch3d.ops.knn_points on the GPU, any point dimension (see the module docstring)."""
    dist, idx = _KnnFunction.apply(p1, p2, lengths1, lengths2, int(K))
    nn = knn_gather(p2, idx, lengths2) if return_nn else None
    return _KNN(dists=dist, idx=idx, knn=nn)
