"""``pytorch_points.network.operations`` -- the part on the hot path: gather_points, ball_query,
grouping_operation, QueryAndGroup (reference: network/operations.py:38-213).  channel_shuffle,
jitter, batch_svd and the torch one-liners of that file are out of scope."""
import torch

from .._ext import sampling


class GatherFunction(torch.autograd.Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint)   (reference operations.py:38-82)"""

    @staticmethod
    def forward(ctx, features, idx):
        features = features.contiguous()
        idx = idx.contiguous()
        idx = idx.to(dtype=torch.int32)

        B, npoint = idx.size()
        _, C, N = features.size()

        output = torch.empty(B, C, npoint, dtype=features.dtype, device=features.device)
        sampling.gather_forward(B, C, N, npoint, features, idx, output)

        ctx.save_for_backward(idx)
        ctx.C = C
        ctx.N = N
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        B, npoint = idx.size()

        grad_features = torch.zeros(B, ctx.C, ctx.N, dtype=grad_out.dtype, device=grad_out.device)
        sampling.gather_backward(B, ctx.C, ctx.N, npoint, grad_out.contiguous(), idx, grad_features)

        return grad_features, None


gather_points = GatherFunction.apply  # type: ignore


class BallQuery(torch.autograd.Function):
    """(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3)) -> int32 (B,npoint,nsample)
    (reference operations.py:88-111; note the argument order)"""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        idx = sampling.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply  # type: ignore


class GroupingOperation(torch.autograd.Function):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)
    (reference operations.py:117-156)"""

    @staticmethod
    def forward(ctx, features, idx):
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()

        ctx.for_backwards = (idx, N)

        return sampling.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards

        grad_features = sampling.group_points_grad(grad_out.contiguous(), idx, N)

        return grad_features, None


grouping_operation = GroupingOperation.apply  # type: ignore


class QueryAndGroup(torch.nn.Module):
    r"""Groups with a ball query of radius (reference operations.py:162-213).

    forward(xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) or None)
    -> (B, 3 + C, npoint, nsample)"""

    def __init__(self, radius, nsample, use_xyz=True):
        super(QueryAndGroup, self).__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
        # the reference subtracts in place on the Function's output (operations.py:197)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)

        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)  # (B, C + 3, npoint, nsample)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz

        return new_features
