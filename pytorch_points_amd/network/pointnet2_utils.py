"""Drop-in for the hot-path names of ``pytorch_points.network.pointnet2_utils``: ThreeNN / three_nn,
ThreeInterpolate / three_interpolate, QueryAndGroup, GroupAll.  Behaviour follows the reference
(network/pointnet2_utils.py); the code is this package's own."""
import torch
from torch import nn

from .._ext import sampling
from .operations import QueryAndGroup, ball_query, grouping_operation  # noqa: F401  (re-exported names)


def _need_contiguous(**tensors):
    for name, t in tensors.items():
        if not t.is_contiguous():
            raise AssertionError("%s must be contiguous" % name)   # the reference asserts (:20-21, :51-53)


class ThreeNN(torch.autograd.Function):
    """``three_nn(unknown (B,n,3), known (B,m,3))`` -> ``(dist (B,n,3), idx (B,n,3) int32)``: Euclidean
    distances (the kernel's squared distances, square-rooted as at reference :33) to the three nearest
    known points, ascending.  Not differentiable (reference :11-37)."""

    @staticmethod
    def forward(ctx, unknown, known):
        _need_contiguous(unknown=unknown, known=known)
        batch, n = unknown.shape[0], unknown.shape[1]
        sq = unknown.new_empty((batch, n, 3), dtype=torch.float32)
        nearest = torch.empty((batch, n, 3), dtype=torch.int32, device=unknown.device)
        sampling.three_nn_wrapper(batch, n, known.shape[1], unknown, known, sq, nearest)
        ctx.mark_non_differentiable(nearest)
        return sq.sqrt_(), nearest

    @staticmethod
    def backward(ctx, *unused):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(torch.autograd.Function):
    """``three_interpolate(features (B,c,m), idx (B,n,3), weight (B,n,3))`` -> ``(B,c,n)``: each output is
    the weighted sum of three input columns; the gradient reaches ``features`` only (reference :43-86)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        _need_contiguous(features=features, idx=idx, weight=weight)
        batch, channels, m = features.shape
        n = idx.shape[1]
        ctx.save_for_backward(idx, weight)
        ctx.source_points = m
        out = features.new_empty((batch, channels, n), dtype=torch.float32)
        sampling.three_interpolate_wrapper(batch, channels, m, n, features, idx, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        batch, channels, n = grad_out.shape
        m = ctx.source_points
        grad_features = grad_out.new_zeros((batch, channels, m), dtype=torch.float32)   # the kernel accumulates
        sampling.three_interpolate_grad_wrapper(batch, channels, n, m, grad_out.detach().contiguous(), idx, weight,
                                                grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupAll(nn.Module):
    """The whole cloud as one group: ``(xyz (B,N,3), new_xyz ignored, features (B,C,N) | None)`` ->
    ``(B, 3 + C, 1, N)`` (coordinates first; without ``use_xyz`` the features alone) -- reference :127-150."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        coords = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return coords
        feats = features.unsqueeze(2)
        return torch.cat((coords, feats), dim=1) if self.use_xyz else feats
