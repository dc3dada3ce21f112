"""Drop-in for the hot-path names of ``pytorch_points.network.geo_operations``: FurthestPointSampling /
furthest_point_sample (reference network/geo_operations.py:11-64).  The mesh-geometry functions of that
file are out of scope."""
import torch

from .._ext import sampling
from .operations import gather_points

_FAR = 1e10   # initial running minimum of every point (reference :29)


class FurthestPointSampling(torch.autograd.Function):
    """``(xyz (B,N,3), npoint, seedIdx)`` -> int32 ``(B,npoint)``: iterative farthest point sampling started
    at point ``seedIdx``.  Not differentiable."""

    @staticmethod
    def forward(ctx, xyz, npoint, seedIdx):
        batch, n = xyz.shape[0], xyz.shape[1]
        picked = torch.empty((batch, npoint), dtype=torch.int32, device=xyz.device)
        # (the reference fills a temp of 1e10 here, :32-33, and never reads it back: temp=None says exactly that)
        sampling.furthest_sampling(npoint, seedIdx, xyz, None, picked)
        ctx.mark_non_differentiable(picked)
        return picked

    @staticmethod
    def backward(ctx, *unused):
        return None, None, None


_furthest_point_sample = FurthestPointSampling.apply  # type: ignore


class _SampleAndGather(torch.autograd.Function):
    """``furthest_point_sample``'s two steps -- the sampling and ``gather_points`` of the coordinates (reference
    :59-63) -- as ONE launch: ``(points (B,N,3), npoint, seedIdx)`` -> ``(idx (B,npoint) int32,
    chosen (B,3,npoint))``.  The gradient of ``chosen`` is scattered back to ``points`` exactly as
    ``gather_points``' backward does (the ordered form under ``torch.use_deterministic_algorithms``)."""

    @staticmethod
    def forward(ctx, points, npoint, seedIdx):
        batch, n = points.shape[0], points.shape[1]
        picked = torch.empty((batch, npoint), dtype=torch.int32, device=points.device)
        chosen = torch.empty((batch, 3, npoint), dtype=torch.float32, device=points.device)
        sampling.furthest_sampling(npoint, seedIdx, points, None, picked, chosen, True)
        ctx.mark_non_differentiable(picked)
        ctx.save_for_backward(picked)
        ctx.n = n
        return picked, chosen

    @staticmethod
    def backward(ctx, _grad_idx, grad_chosen):
        (picked,) = ctx.saved_tensors
        if grad_chosen is None:
            return None, None, None
        batch, npoint = picked.shape
        grad_cf = grad_chosen.new_zeros((batch, 3, ctx.n))          # the kernel accumulates
        sampling.gather_backward(batch, 3, ctx.n, npoint, grad_chosen.contiguous(), picked, grad_cf)
        return grad_cf.transpose(2, 1), None, None


def furthest_point_sample(xyz, npoint, NCHW=True, seedIdx=0):
    """Sample ``npoint`` points; returns ``(idx (B,npoint) int32, points)`` with ``points`` in the layout of
    the input: ``(B,3,npoint)`` for ``NCHW`` input ``(B,3,N)``, ``(B,npoint,3)`` for ``(B,N,3)``
    (reference :44-64, same messages)."""
    assert (xyz.dim() == 3), "input for furthest sampling must be a 3D-tensor, but xyz.size() is {}".format(xyz.size())
    points_last = xyz.transpose(2, 1) if NCHW else xyz          # (B, N, 3) view
    assert (points_last.size(2) == 3), "furthest sampling is implemented for 3D points"
    points_last = points_last.contiguous()
    if points_last.dtype is torch.float32 and npoint >= 1:
        idx, chosen = _SampleAndGather.apply(points_last, npoint, seedIdx)       # one launch; (B, 3, npoint)
    else:
        idx = _furthest_point_sample(points_last, npoint, seedIdx)
        chosen = gather_points(points_last.transpose(2, 1).contiguous(), idx)   # (B, 3, npoint)
    return idx, (chosen if NCHW else chosen.transpose(2, 1).contiguous())
