"""``pytorch_points.network.pointnet2_utils`` (reference: network/pointnet2_utils.py): ThreeNN /
three_nn, ThreeInterpolate / three_interpolate, QueryAndGroup, GroupAll."""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from .._ext import sampling
from .operations import grouping_operation, ball_query, QueryAndGroup  # noqa: F401  (twin class, :91-124)


class ThreeNN(Function):
    """unknown (B,N,3), known (B,M,3) -> (dist (B,N,3) L2 distances ascending, idx (B,N,3) int32)
    (reference pointnet2_utils.py:11-37; returns sqrt of the kernel's squared distances, :33)"""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()

        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, N, 3, dtype=torch.int32, device=unknown.device)

        sampling.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n)
    (reference pointnet2_utils.py:43-86)"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()

        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty(B, c, n, dtype=torch.float32, device=features.device)

        sampling.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()

        grad_features = torch.zeros(B, c, m, dtype=torch.float32, device=grad_out.device)
        grad_out_data = grad_out.data.contiguous()

        sampling.three_interpolate_grad_wrapper(B, c, n, m, grad_out_data, idx, weight, grad_features.data)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupAll(nn.Module):
    """(reference pointnet2_utils.py:127-150) -> (B, C + 3, 1, N)"""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)  # (B, 3 + C, 1, N)
            else:
                new_features = grouped_features
        else:
            new_features = grouped_xyz

        return new_features
