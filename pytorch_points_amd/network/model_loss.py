"""Drop-in for the hot-path names of ``pytorch_points.network.model_loss``: NmDistanceFunction /
nndistance and LabeledNmdistanceFunction / labeled_nndistance (reference network/model_loss.py:401-483).
The twelve torch-composed loss modules of that file are out of scope."""
import torch

from .._ext import losses


def _outputs(xyz1, xyz2):
    """(dist1 (B,N), dist2 (B,M), idx1, idx2) uninitialised, on the inputs' device.  The kernels write
    every element (and zero-fill by themselves when one cloud is empty, the only case in which the
    reference's zero initialisation survives), so no fill launches are spent.  The reference allocates on
    the CPU and moves to the *current* device (:412-421)."""
    n, m = xyz1.shape[1], xyz2.shape[1]
    batch = xyz1.shape[0]
    return (xyz1.new_empty((batch, n)), xyz1.new_empty((batch, m)),
            torch.empty((batch, n), dtype=torch.int32, device=xyz1.device),
            torch.empty((batch, m), dtype=torch.int32, device=xyz1.device))


def _finish_forward(ctx, xyz1, xyz2, dist1, dist2, idx1, idx2):
    ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
    ctx.mark_non_differentiable(idx1, idx2)
    ctx.set_materialize_grads(False)   # no zero tensors for the two index outputs on every backward
    return dist1, dist2, idx1, idx2


def _chamfer_backward(ctx, grad1, grad2):
    """d dist1[i] / d xyz1[i] = 2 (xyz1[i] - xyz2[idx1[i]]) and the mirrored terms (reference kernel
    nmdistance_cuda.cu:168-185); a missing upstream gradient counts as zero."""
    xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
    grad1 = xyz1.new_zeros(idx1.shape) if grad1 is None else grad1.contiguous()
    grad2 = xyz2.new_zeros(idx2.shape) if grad2 is None else grad2.contiguous()
    out1, out2 = torch.empty_like(xyz1), torch.empty_like(xyz2)      # fully overwritten by the kernel
    losses.nmdistance_backward(xyz1, xyz2, out1, out2, grad1, grad2, idx1, idx2)
    return out1, out2


class NmDistanceFunction(torch.autograd.Function):
    """``nndistance(xyz1 (B,N,C), xyz2 (B,M,C))`` -> ``(dist1 (B,N), dist2 (B,M), idx1, idx2)``: squared
    distance from every point to its nearest neighbour in the other cloud, and that neighbour's int32 index
    (not differentiable).  Reference :401-439."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        assert xyz1.dtype == xyz2.dtype
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        out = _outputs(xyz1, xyz2)
        losses.nmdistance_forward(xyz1, xyz2, *out)
        return _finish_forward(ctx, xyz1, xyz2, *out)

    @staticmethod
    def backward(ctx, graddist1, graddist2, *index_grads):
        return _chamfer_backward(ctx, graddist1, graddist2)


nndistance = NmDistanceFunction.apply  # type: ignore


class LabeledNmdistanceFunction(torch.autograd.Function):
    """``labeled_nndistance(xyz1, xyz2, label1 (B,N), label2 (B,M))``: nearest neighbour among the points
    of the other cloud that carry the same label; a point without such a partner gets idx -1, dist 0
    (reference :445-481).  The inputs are made contiguous here (the reference omits it);
    labels are compared in the coordinates' dtype, as the reference's kernel does."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, label1, label2):
        assert xyz1.dtype == xyz2.dtype
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        out = _outputs(xyz1, xyz2)
        losses.labeled_nmdistance_forward(xyz1, xyz2, label1.to(xyz1.dtype), label2.to(xyz1.dtype), *out)
        return _finish_forward(ctx, xyz1, xyz2, *out)

    @staticmethod
    def backward(ctx, graddist1, graddist2, *index_grads):
        return _chamfer_backward(ctx, graddist1, graddist2) + (None, None)


labeled_nndistance = LabeledNmdistanceFunction.apply
