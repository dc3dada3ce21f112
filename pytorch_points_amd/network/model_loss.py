"""``pytorch_points.network.model_loss`` -- the part on the hot path: NmDistanceFunction /
nndistance and LabeledNmdistanceFunction / labeled_nndistance
(reference: network/model_loss.py:401-483).  The twelve torch-composed loss modules of that file
are out of scope."""
import torch

from .._ext import losses


class NmDistanceFunction(torch.autograd.Function):
    """3D point set to 3D point set distance (reference network/model_loss.py:401-439).

    Returns ``(dist1, dist2, idx1, idx2)``: squared distance from every xyz1 point to its nearest
    xyz2 point and the converse, with the int32 indices of those neighbours (non-differentiable).
    Outputs live on the inputs' device (the reference allocates on the CPU and ``.cuda()``s them to
    the current device, :412-421)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert xyz1.dtype == xyz2.dtype
        # the kernel overwrites every element (and zero-fills them itself when n == 0 or m == 0, the
        # only case in which the reference's zeros survive), so no fill launches are needed
        dist1 = torch.empty(batchsize, n, dtype=xyz1.dtype, device=xyz1.device)
        dist2 = torch.empty(batchsize, m, dtype=xyz1.dtype, device=xyz1.device)
        idx1 = torch.empty(batchsize, n, dtype=torch.int32, device=xyz1.device)
        idx2 = torch.empty(batchsize, m, dtype=torch.int32, device=xyz1.device)
        losses.nmdistance_forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        # do not let autograd zero-fill gradient tensors for the two index outputs on every backward
        ctx.set_materialize_grads(False)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradNone1, gradNone2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = torch.zeros_like(idx1, dtype=xyz1.dtype) if graddist1 is None else graddist1.contiguous()
        graddist2 = torch.zeros_like(idx2, dtype=xyz2.dtype) if graddist2 is None else graddist2.contiguous()
        gradxyz1 = torch.empty_like(xyz1)  # fully overwritten by the kernel
        gradxyz2 = torch.empty_like(xyz2)
        losses.nmdistance_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        return gradxyz1, gradxyz2


nndistance = NmDistanceFunction.apply  # type: ignore


class LabeledNmdistanceFunction(torch.autograd.Function):
    """CD within the same category; points with no same-label partner get idx -1, dist 0
    (reference network/model_loss.py:445-481).  Inputs are made contiguous here (the reference
    omits it -- a latent bug)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, label1, label2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        assert xyz1.dtype == xyz2.dtype
        label1 = label1.to(dtype=xyz1.dtype)
        label2 = label2.to(dtype=xyz1.dtype)
        dist1 = torch.empty(batchsize, n, dtype=xyz1.dtype, device=xyz1.device)
        dist2 = torch.empty(batchsize, m, dtype=xyz1.dtype, device=xyz1.device)
        idx1 = torch.empty(batchsize, n, dtype=torch.int32, device=xyz1.device)
        idx2 = torch.empty(batchsize, m, dtype=torch.int32, device=xyz1.device)
        losses.labeled_nmdistance_forward(xyz1, xyz2, label1, label2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        ctx.set_materialize_grads(False)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradNone1, gradNone2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = torch.zeros_like(idx1, dtype=xyz1.dtype) if graddist1 is None else graddist1.contiguous()
        graddist2 = torch.zeros_like(idx2, dtype=xyz2.dtype) if graddist2 is None else graddist2.contiguous()
        gradxyz1 = torch.empty_like(xyz1)
        gradxyz2 = torch.empty_like(xyz2)
        losses.nmdistance_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        return gradxyz1, gradxyz2, None, None


labeled_nndistance = LabeledNmdistanceFunction.apply
