"""``pytorch_points.network.operations`` -- the part on the hot path: gather_points, ball_query,
grouping_operation, QueryAndGroup (reference: network/operations.py:38-213).  channel_shuffle,
jitter, batch_svd and the torch one-liners of that file are out of scope."""
import torch

from .._ext import sampling


class GatherFunction(torch.autograd.Function):
    """``gather_points(features (B,C,N), idx (B,npoint))`` -> ``(B,C,npoint)``: columns of ``features``
    picked by ``idx`` (int32; other integer types are converted, reference operations.py:55); the
    gradient is scattered back to ``features`` (reference :38-82)."""

    @staticmethod
    def forward(ctx, features, idx):
        src = features.contiguous()
        cols = idx.contiguous().to(torch.int32)
        batch, channels, n = src.shape
        picked = cols.shape[1]
        out = src.new_empty((batch, channels, picked))
        sampling.gather_forward(batch, channels, n, picked, src, cols, out)
        ctx.save_for_backward(cols)
        ctx.source_shape = (channels, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (cols,) = ctx.saved_tensors
        channels, n = ctx.source_shape
        batch, picked = cols.shape
        grad_features = grad_out.new_zeros((batch, channels, n))      # the kernel accumulates
        sampling.gather_backward(batch, channels, n, picked, grad_out.contiguous(), cols, grad_features)
        return grad_features, None


gather_points = GatherFunction.apply  # type: ignore


class BallQuery(torch.autograd.Function):
    """``ball_query(radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3))`` -> int32 ``(B,npoint,nsample)``:
    per centre the first ``nsample`` points of ``xyz`` (in index order) closer than ``radius``, padded with
    the first hit.  Argument order as the reference's (operations.py:88-111).  Not differentiable."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        members = sampling.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(members)
        return members

    @staticmethod
    def backward(ctx, *unused):
        return None, None, None, None


ball_query = BallQuery.apply  # type: ignore


class GroupingOperation(torch.autograd.Function):
    """``grouping_operation(features (B,C,N), idx (B,npoint,nsample))`` -> ``(B,C,npoint,nsample)``; the
    gradient is scattered back to ``features`` (reference operations.py:117-156)."""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.source_points = features.shape[2]
        return sampling.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return sampling.group_points_grad(grad_out.contiguous(), idx, ctx.source_points), None


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
        """The same result composed from the public operators the way the reference does it
        (operations.py:193-204: ball_query, two grouping_operations, a subtraction and a concatenation);
        kept as the yardstick of the parity tests and of ``bench.py --workload ball_group``."""
        if features is None and not self.use_xyz:
            raise AssertionError("Cannot have not features and not use xyz as a feature!")
        members = ball_query(self.radius, self.nsample, xyz, new_xyz)
        parts = []
        if self.use_xyz or features is None:
            local = grouping_operation(xyz.transpose(1, 2).contiguous(), members)
            parts.append(local - new_xyz.transpose(1, 2).unsqueeze(-1))     # relative to the centre
        if features is not None:
            parts.append(grouping_operation(features, members))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
