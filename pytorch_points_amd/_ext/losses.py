"""``pytorch_points._ext.losses`` (reference: _ext/nmdistance.cpp:30-34).

Same names, same positional arguments, same ``int`` return (1 = ok, as chamfer_cuda_forward
returns on success, _ext/nmdistance_cuda.cu:137).  Unlike the reference, which checks nothing
here, inputs are validated and failures raise RuntimeError.
"""
import os

import torch

from .. import _lib

# PP_NMDISTANCE_SEARCH=bruteforce (read once, at import): every pair evaluated, no grid search
_FORCE_BRUTE = os.environ.get("PP_NMDISTANCE_SEARCH") == "bruteforce"


def _workspace_bytes(L, b, n, m, c, labeled):
    """bytes of scratch the grid search wants for this shape (0: the brute force serves it)"""
    if _FORCE_BRUTE:
        return 0
    if labeled:
        return L.pp_labeled_nmdistance_forward_workspace_bytes(b, n, m, c)
    return L.pp_nmdistance_forward_workspace_bytes(b, n, m, c)


def _shapes(xyz1, xyz2):
    if xyz1.dim() != 3 or xyz2.dim() != 3:
        raise RuntimeError("xyz1 and xyz2 must be (B, N, C) and (B, M, C)")
    b, n, c = xyz1.shape
    b2, m, c2 = xyz2.shape
    if b != b2 or c != c2:  # CHECK_EQ(xyz2.size(2), c), _ext/nmdistance_cuda.cu:124
        raise RuntimeError("xyz1 %s and xyz2 %s disagree in batch or point dimension"
                           % (tuple(xyz1.shape), tuple(xyz2.shape)))
    return b, n, m, c


def nmdistance_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_forward (_ext/nmdistance.cpp:13-15): fills dist1/idx1 (B,N), dist2/idx2 (B,M)."""
    b, n, m, c = _shapes(xyz1, xyz2)
    floats = (("xyz1", xyz1), ("xyz2", xyz2), ("dist1", dist1), ("dist2", dist2))
    ints = (("idx1", idx1), ("idx2", idx2))
    dev = _lib.require_cuda(*floats, *ints)
    _lib.require_contiguous(*floats, *ints)
    dt = _lib.require_float_or_double(*floats)
    _lib.require_int(*ints)
    if dist1.numel() != b * n or idx1.numel() != b * n or dist2.numel() != b * m or idx2.numel() != b * m:
        raise RuntimeError("output tensors must be (B, N) and (B, M)")
    if dt is not torch.float32:
        _launch_f64("pp_nmdistance_forward_" + _SUFFIX[dt], "nmdistance_forward", dev,
                    xyz1, xyz2, dist1, idx1, dist2, idx2, b, n, m, c)
    else:
        _launch_forward(xyz1, xyz2, dist1, dist2, idx1, idx2, b, n, m, c, dev)
    return 1


_SUFFIX = {torch.float64: "f64", torch.float16: "f16"}   # the every-pair scans for the reference's other two types


def _launch_f64(symbol, what, dev, *args):
    """double / half clouds (the reference's scalar_t = double / at::Half instantiations, _ext/nmdistance_cuda.cu:125,210): tensors
    first (passed by address), then the four sizes; arguments already validated"""
    fn = getattr(_lib.lib(), symbol)
    with _lib.on_device(dev) as stream:
        code = fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args], stream)
    if code:
        _lib.check(code, what)


def _launch_forward(xyz1, xyz2, dist1, dist2, idx1, idx2, b, n, m, c, dev):
    """the launch itself; arguments already validated (the autograd operator validates its two inputs and
    allocates the four outputs itself, so it comes straight here)"""
    L = _lib.lib()
    nbytes = _workspace_bytes(L, b, n, m, c, False)
    ws = _lib.workspace(dev, "nmdistance", nbytes)
    idx = dev.index
    if idx is not None and idx != _lib.current_device():
        with torch.cuda.device(dev):
            code = L.pp_nmdistance_forward_ws_f32(
                xyz1.data_ptr(), xyz2.data_ptr(), dist1.data_ptr(), idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(),
                b, n, m, c, ws.data_ptr() if ws is not None else None, nbytes, _lib.raw_stream(dev))
    else:
        code = L.pp_nmdistance_forward_ws_f32(
            xyz1.data_ptr(), xyz2.data_ptr(), dist1.data_ptr(), idx1.data_ptr(), dist2.data_ptr(), idx2.data_ptr(),
            b, n, m, c, ws.data_ptr() if ws is not None else None, nbytes, _lib.raw_stream(dev))
    if code:
        _lib.check(code, "nmdistance_forward")


def labeled_nmdistance_forward(xyz1, xyz2, label1, label2, dist1, dist2, idx1, idx2):
    """labeled_chamfer_forward (_ext/nmdistance.cpp:17-20).  Labels are converted to the xyz dtype
    as the reference does (``label.toType(xyz1.scalar_type())``, _ext/nmdistance_cuda.cu:153)."""
    b, n, m, c = _shapes(xyz1, xyz2)
    label1 = label1.to(dtype=xyz1.dtype).contiguous()
    label2 = label2.to(dtype=xyz1.dtype).contiguous()
    floats = (("xyz1", xyz1), ("xyz2", xyz2), ("label1", label1), ("label2", label2),
              ("dist1", dist1), ("dist2", dist2))
    ints = (("idx1", idx1), ("idx2", idx2))
    dev = _lib.require_cuda(*floats, *ints)
    _lib.require_contiguous(*floats, *ints)
    _lib.require_float(*floats)
    _lib.require_int(*ints)
    if label1.numel() != b * n or label2.numel() != b * m:
        raise RuntimeError("labels must be (B, N) and (B, M)")
    L = _lib.lib()
    nbytes = _workspace_bytes(L, b, n, m, c, True)
    ws = _lib.workspace(dev, "nmdistance", nbytes)
    with _lib.on_device(dev) as stream:
        _lib.check(L.pp_labeled_nmdistance_forward_ws_f32(
            _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(label1), _lib.ptr(label2), _lib.ptr(dist1),
            _lib.ptr(idx1), _lib.ptr(dist2), _lib.ptr(idx2), b, n, m, c,
            _lib.ptr(ws) if ws is not None else None, nbytes, stream), "labeled_nmdistance_forward")
    return 1


def nmdistance_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_backward (_ext/nmdistance.cpp:23-27): overwrites gradxyz1 (B,N,C), gradxyz2 (B,M,C)."""
    b, n, m, c = _shapes(xyz1, xyz2)
    floats = (("xyz1", xyz1), ("xyz2", xyz2), ("gradxyz1", gradxyz1), ("gradxyz2", gradxyz2),
              ("graddist1", graddist1), ("graddist2", graddist2))
    ints = (("idx1", idx1), ("idx2", idx2))
    dev = _lib.require_cuda(*floats, *ints)
    _lib.require_contiguous(*floats, *ints)
    dt = _lib.require_float_or_double(*floats)
    _lib.require_int(*ints)
    if gradxyz1.shape != xyz1.shape or gradxyz2.shape != xyz2.shape:
        raise RuntimeError("gradxyz tensors must have the shapes of xyz1 / xyz2")
    if dt is not torch.float32:
        _launch_f64("pp_nmdistance_backward_" + _SUFFIX[dt], "nmdistance_backward", dev,
                    xyz1, xyz2, graddist1, graddist2, idx1, idx2, gradxyz1, gradxyz2, b, n, m, c)
    else:
        _launch_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2, b, n, m, c, dev)
    return 1


def _launch_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2, b, n, m, c, dev):
    """the launch itself; arguments already validated.  torch.use_deterministic_algorithms(True) selects the
    ordered form (ascending source order, no floating-point atomics: include/pp_hip.h)"""
    L = _lib.lib()
    if _lib.deterministic():
        with _lib.on_device(dev) as stream:
            if _lib.ordered_or_fallback(L.pp_nmdistance_backward_ordered_f32(
                    xyz1.data_ptr(), xyz2.data_ptr(), graddist1.data_ptr(), graddist2.data_ptr(), idx1.data_ptr(),
                    idx2.data_ptr(), gradxyz1.data_ptr(), gradxyz2.data_ptr(), b, n, m, c, stream), "nmdistance_backward"):
                return
    idx = dev.index
    if idx is not None and idx != _lib.current_device():
        with torch.cuda.device(dev):
            code = L.pp_nmdistance_backward_f32(
                xyz1.data_ptr(), xyz2.data_ptr(), graddist1.data_ptr(), graddist2.data_ptr(), idx1.data_ptr(),
                idx2.data_ptr(), gradxyz1.data_ptr(), gradxyz2.data_ptr(), b, n, m, c, _lib.raw_stream(dev))
    else:
        code = L.pp_nmdistance_backward_f32(
            xyz1.data_ptr(), xyz2.data_ptr(), graddist1.data_ptr(), graddist2.data_ptr(), idx1.data_ptr(),
            idx2.data_ptr(), gradxyz1.data_ptr(), gradxyz2.data_ptr(), b, n, m, c, _lib.raw_stream(dev))
    if code:
        _lib.check(code, "nmdistance_backward")
