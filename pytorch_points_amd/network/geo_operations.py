"""``pytorch_points.network.geo_operations`` -- the part on the hot path: FurthestPointSampling /
furthest_point_sample (reference: network/geo_operations.py:11-64).  The mesh-geometry functions of
that file are out of scope."""
import torch

from .._ext import sampling
from .operations import gather_points


class FurthestPointSampling(torch.autograd.Function):
    """xyz (B,N,3), npoint, seedIdx -> int32 idx (B,npoint)   (reference geo_operations.py:11-38)"""

    @staticmethod
    def forward(ctx, xyz, npoint, seedIdx):
        B, N, _ = xyz.size()

        idx = torch.empty([B, npoint], dtype=torch.int32, device=xyz.device)
        temp = torch.full([B, N], 1e10, dtype=torch.float32, device=xyz.device)
        sampling.furthest_sampling(npoint, seedIdx, xyz, temp, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_idx=None):
        return None, None, None


_furthest_point_sample = FurthestPointSampling.apply  # type: ignore


def furthest_point_sample(xyz, npoint, NCHW=True, seedIdx=0):
    """
    :param
        xyz (B, 3, N) or (B, N, 3)
        npoint a constant
    :return
        torch.IntTensor
            (B, npoint) tensor containing the indices
        torch.FloatTensor
            (B, npoint, 3) or (B, 3, npoint) point sets
    (reference geo_operations.py:44-64)"""
    assert (xyz.dim() == 3), "input for furthest sampling must be a 3D-tensor, but xyz.size() is {}".format(xyz.size())
    # need transpose
    if NCHW:
        xyz = xyz.transpose(2, 1).contiguous()

    assert (xyz.size(2) == 3), "furthest sampling is implemented for 3D points"
    idx = _furthest_point_sample(xyz.contiguous(), npoint, seedIdx)
    sampled_pc = gather_points(xyz.transpose(2, 1).contiguous(), idx)
    if not NCHW:
        sampled_pc = sampled_pc.transpose(2, 1).contiguous()
    return idx, sampled_pc
