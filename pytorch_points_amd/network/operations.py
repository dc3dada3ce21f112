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


class _QueryAndGroupFused(torch.autograd.Function):
    """ball_query -> group(xyz) - centre -> group(features) -> concat as ONE output tensor written
    once.  The reference composes four ops and a torch.cat (operations.py:193-204); at B=32, C=128,
    npoint=4096, nsample=64 the cat alone re-reads and re-writes 4.4 GB.  Values are identical."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, radius, nsample, use_xyz):
        xyz = xyz.contiguous()
        new_xyz = new_xyz.contiguous()
        idx = sampling.ball_query(new_xyz, xyz, radius, nsample)
        B, N, _ = xyz.shape
        npoint = new_xyz.shape[1]
        C = features.shape[1] if features is not None else 0
        cx = 3 if use_xyz else 0
        out = torch.empty(B, cx + C, npoint, nsample, dtype=torch.float32, device=xyz.device)
        if use_xyz:
            sampling.group_points_into(xyz.transpose(1, 2).contiguous(), idx, out, 0)
            out[:, :3] -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            sampling.group_points_into(features.contiguous(), idx, out, cx)
        ctx.for_backwards = (idx, N, C, cx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N, C, cx = ctx.for_backwards
        grad_out = grad_out.contiguous()
        grad_xyz = grad_new_xyz = grad_features = None
        if cx and ctx.needs_input_grad[0]:
            grad_xyz = sampling.group_points_grad_from(grad_out, idx, N, 0, 3).transpose(1, 2).contiguous()
        if cx and ctx.needs_input_grad[1]:
            grad_new_xyz = -grad_out[:, :3].sum(-1).transpose(1, 2).contiguous()
        if C and ctx.needs_input_grad[2]:
            grad_features = sampling.group_points_grad_from(grad_out, idx, N, cx, C)
        return grad_xyz, grad_new_xyz, grad_features, None, None, None


class QueryAndGroup(torch.nn.Module):
    r"""Groups with a ball query of radius (reference operations.py:162-213).

    forward(xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) or None)
    -> (B, 3 + C, npoint, nsample)"""

    def __init__(self, radius, nsample, use_xyz=True):
        super(QueryAndGroup, self).__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return _QueryAndGroupFused.apply(xyz, new_xyz, features, self.radius, self.nsample, self.use_xyz)

    def forward_unfused(self, xyz, new_xyz, features=None):
        """The reference's composition, op by op (operations.py:193-204); kept for the parity tests."""
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, nsample)
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
