"""Drop-in for the hot-path names of ``pytorch_points.network.model_loss``: NmDistanceFunction /
nndistance and LabeledNmdistanceFunction / labeled_nndistance (reference network/model_loss.py:401-483).
The twelve torch-composed loss modules of that file are out of scope.

``nndistance`` / ``labeled_nndistance`` are the C++ autograd nodes of csrc/torch_bridge.cpp (the reference's
host side is a C++ extension too): at config 2 the step's kernels take less time than Python needs to issue
them through ``torch.autograd.Function``, and the autograd engine has to take the GIL on its device thread for
a Python backward.  The Python classes below are the same operators over the same C ABI (the reference's class
names; ``NmDistanceFunction.apply`` works as in the reference) and are tested to agree bit for bit."""
import torch

from .. import _lib
from .._ext import losses


def _inputs(xyz1, xyz2, double_ok=False):
    """contiguous fp32 (double_ok: or fp64) GPU clouds (B,N,C), (B,M,C) on one device -> (xyz1, xyz2, B, N, M, C,
    device); the checks the reference leaves out (its launcher validates nothing, _ext/nmdistance.cpp:13-15)"""
    assert xyz1.dtype == xyz2.dtype
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    if xyz1.dtype is torch.bfloat16:
        raise TypeError("xyz1 is %s: the Chamfer operators serve float32, float64 and float16 (the reference's "
                        "AT_DISPATCH_FLOATING_TYPES_AND_HALF)" % xyz1.dtype)
    if xyz1.dtype is not torch.float32 and not (double_ok and xyz1.dtype in (torch.float64, torch.float16)):
        if xyz1.dtype in (torch.float64, torch.float16):
            raise TypeError("xyz1 is %s: this operator serves float32 only" % xyz1.dtype)
        raise RuntimeError("xyz1 must be a float tensor")
    if not (xyz1.is_cuda and xyz2.is_cuda):
        raise RuntimeError("%s must be a CUDA tensor" % ("xyz2" if xyz1.is_cuda else "xyz1"))
    dev = xyz1.device
    if xyz2.device != dev:
        raise RuntimeError("xyz2 is on %s, expected %s" % (xyz2.device, dev))
    b, n, m, c = losses._shapes(xyz1, xyz2)
    return xyz1, xyz2, b, n, m, c, dev


def _outputs(b, n, m, dev, dtype=torch.float32):
    """(dist1 (B,N), dist2 (B,M), idx1, idx2) uninitialised, on the inputs' device.  The kernels write
    every element (and zero-fill by themselves when one cloud is empty, the only case in which the
    reference's zero initialisation survives), so no fill launches are spent.  The reference allocates on
    the CPU and moves to the *current* device (:412-421)."""
    return (torch.empty((b, n), dtype=dtype, device=dev), torch.empty((b, m), dtype=dtype, device=dev),
            torch.empty((b, n), dtype=torch.int32, device=dev), torch.empty((b, m), dtype=torch.int32, device=dev))


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
    if grad1.dtype is not xyz1.dtype or grad2.dtype is not xyz1.dtype:
        raise RuntimeError("graddist1 must be a %s tensor" % {torch.float64: "double", torch.float16: "half"}.get(xyz1.dtype, "float"))
    dev = xyz1.device
    if grad1.device != dev or grad2.device != dev:
        raise RuntimeError("graddist is on another device than xyz1 (%s)" % (dev,))
    out1, out2 = torch.empty_like(xyz1), torch.empty_like(xyz2)      # fully overwritten by the kernel
    b, n, c = xyz1.shape
    if xyz1.dtype is not torch.float32:
        losses._launch_f64("pp_nmdistance_backward_" + losses._SUFFIX[xyz1.dtype], "nmdistance_backward", dev,
                           xyz1, xyz2, grad1, grad2, idx1, idx2, out1, out2, b, n, xyz2.shape[1], c)
    else:
        losses._launch_backward(xyz1, xyz2, out1, out2, grad1, grad2, idx1, idx2, b, n, xyz2.shape[1], c, dev)
    return out1, out2


class NmDistanceFunction(torch.autograd.Function):
    """``nndistance(xyz1 (B,N,C), xyz2 (B,M,C))`` -> ``(dist1 (B,N), dist2 (B,M), idx1, idx2)``: squared
    distance from every point to its nearest neighbour in the other cloud, and that neighbour's int32 index
    (not differentiable).  Reference :401-439."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2, b, n, m, c, dev = _inputs(xyz1, xyz2, double_ok=True)
        out = _outputs(b, n, m, dev, xyz1.dtype)
        if xyz1.dtype is not torch.float32:   # the reference's scalar_t = double / at::Half instantiations (nmdistance_cuda.cu:125)
            losses._launch_f64("pp_nmdistance_forward_" + losses._SUFFIX[xyz1.dtype], "nmdistance_forward", dev,
                               xyz1, xyz2, out[0], out[2], out[1], out[3], b, n, m, c)
        else:
            losses._launch_forward(xyz1, xyz2, *out, b, n, m, c, dev)
        return _finish_forward(ctx, xyz1, xyz2, *out)

    @staticmethod
    def backward(ctx, graddist1, graddist2, *index_grads):
        return _chamfer_backward(ctx, graddist1, graddist2)


def nndistance(xyz1, xyz2):
    """``nndistance(xyz1 (B,N,C), xyz2 (B,M,C))`` -> ``(dist1 (B,N), dist2 (B,M), idx1, idx2)`` (reference :442:
    ``nndistance = NmDistanceFunction.apply``), through the native autograd node.  fp32 is the tuned path; double
    and half clouds (the reference dispatches over the floating types and half, _ext/nmdistance_cuda.cu:125) go through
    the Python node to every-pair kernels with the reference's arithmetic for the type; bfloat16 raises TypeError."""
    if xyz1.dtype is not torch.float32 and xyz1.dtype in (torch.float64, torch.float16, torch.bfloat16):
        return NmDistanceFunction.apply(xyz1, xyz2)
    return _lib.bridge().nndistance(xyz1, xyz2)


class LabeledNmdistanceFunction(torch.autograd.Function):
    """``labeled_nndistance(xyz1, xyz2, label1 (B,N), label2 (B,M))``: nearest neighbour among the points
    of the other cloud that carry the same label; a point without such a partner gets idx -1, dist 0
    (reference :445-481).  The inputs are made contiguous here (the reference omits it);
    labels are compared in the coordinates' dtype, as the reference's kernel does."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, label1, label2):
        xyz1, xyz2, b, n, m, c, dev = _inputs(xyz1, xyz2)
        out = _outputs(b, n, m, dev)
        losses.labeled_nmdistance_forward(xyz1, xyz2, label1.to(xyz1.dtype), label2.to(xyz1.dtype), *out)
        return _finish_forward(ctx, xyz1, xyz2, *out)

    @staticmethod
    def backward(ctx, graddist1, graddist2, *index_grads):
        return _chamfer_backward(ctx, graddist1, graddist2) + (None, None)


def labeled_nndistance(xyz1, xyz2, label1, label2):
    """``labeled_nndistance(xyz1, xyz2, label1 (B,N), label2 (B,M))`` (reference :483), through the native
    autograd node."""
    return _lib.bridge().labeled_nndistance(xyz1, xyz2, label1, label2)
