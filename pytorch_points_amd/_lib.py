"""ctypes binding of libpp_hip.so (C ABI: include/pp_hip.h).

There is NO fallback: if the HIP library is missing or an entry point is absent, importing an
operator raises.  Tensors are passed as raw device pointers, the stream as torch's current HIP
stream on the tensor's device, under a device guard (the reference launches Chamfer and FPS on the
legacy default stream with no guard -- SURVEY.md F10; this is a deliberate correction).
"""
import collections
import ctypes
import os

import torch

from . import _build

_c_void_p, _c_int, _c_float, _c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_P, _I, _F = _c_void_p, _c_int, _c_float

# name -> argtypes (return type is int unless listed in _RESTYPES); must match include/pp_hip.h
SIGNATURES = {
    "pp_version": [],
    "pp_opt_n_threads": [_I],
    "pp_nmdistance_forward_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_nmdistance_forward_workspace_bytes": [_I, _I, _I, _I],
    "pp_nmdistance_forward_ws_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_labeled_nmdistance_forward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_labeled_nmdistance_forward_workspace_bytes": [_I, _I, _I, _I],
    "pp_labeled_nmdistance_forward_ws_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_nmdistance_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_nmdistance_forward_f64": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_nmdistance_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_nmdistance_forward_f16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_nmdistance_backward_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_furthest_sampling_workspace_bytes": [_I, _I, _I],
    "pp_furthest_sampling_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_furthest_sampling_gather_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_furthest_sampling_status": [_P, _P],
    "pp_gather_forward_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "pp_gather_backward_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "pp_ball_query_f32": [_P, _P, _P, _I, _I, _I, _F, _I, _P],
    "pp_ball_query_workspace_bytes": [_I, _I, _I, _I],
    "pp_ball_query_ws_f32": [_P, _P, _P, _I, _I, _I, _F, _I, _P, _c_size_t, _P],
    "pp_shard_packed_bytes": [ctypes.c_longlong, ctypes.c_longlong, _I],
    "pp_shard_pack_f32": [_P, _P, _P, _P, _P, ctypes.c_longlong, ctypes.c_longlong, _I, _P],
    "pp_shard_unpack_f32": [_P, _I, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _I, _P, _P, _P, _P, _P],
    "pp_knn_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_knn_nd_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pp_knn_workspace_bytes": [_I, _I, _I, _I],
    "pp_knn_ws_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_three_nn_workspace_bytes": [_I, _I, _I],
    "pp_three_nn_ws_f32": [_P, _P, _P, _P, _I, _I, _I, _P, _c_size_t, _P],
    "pp_group_points_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pp_group_points_grad_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pp_group_points_strided_f32": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_longlong, _P],
    "pp_group_points_grad_strided_f32": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_longlong, _P],
    "pp_scatter_workspace_bytes": [_I, ctypes.c_longlong, _I, _I, _I],
    "pp_group_points_grad_ws_f32": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_longlong, _P, _c_size_t, _P],
    "pp_group_points_grad_out_ws_f32": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_longlong, _P, _c_size_t, _P],
    "pp_gather_backward_ws_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_three_interpolate_grad_ws_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_nmdistance_backward_ordered_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_group_points_grad_ordered_f32": [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_longlong, _P, _c_size_t, _P],
    "pp_gather_backward_ordered_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_three_interpolate_grad_ordered_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _c_size_t, _P],
    "pp_three_nn_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
    "pp_three_interpolate_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "pp_three_interpolate_grad_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
}
_RESTYPES = {"pp_version": ctypes.c_char_p, "pp_furthest_sampling_workspace_bytes": _c_size_t,
             "pp_nmdistance_forward_workspace_bytes": _c_size_t,
             "pp_labeled_nmdistance_forward_workspace_bytes": _c_size_t,
             "pp_scatter_workspace_bytes": _c_size_t, "pp_ball_query_workspace_bytes": _c_size_t,
             "pp_three_nn_workspace_bytes": _c_size_t, "pp_knn_workspace_bytes": _c_size_t,
             "pp_shard_packed_bytes": _c_size_t}

_lib = None


def library_path():
    return _build.LIB


def lib():
    """The loaded library.  Raises RuntimeError if libpp_hip.so is absent (no CPU fallback)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "pytorch_points_amd: %s not found. Build it with `python -m pytorch_points_amd._build` "
                "(hipcc, gfx950). There is no CPU fallback." % path)
        handle = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _c_int)
        _lib = handle
    return _lib


def version():
    return lib().pp_version().decode()


_bridge = None


def bridge():
    """The C++ autograd nodes (csrc/torch_bridge.cpp -> _pp_torch.so).  Raises RuntimeError when the module has
    not been built: like the kernels, the operators' native host side has no substitute that is picked silently."""
    global _bridge
    if _bridge is None:
        lib()   # the C-ABI library first: the bridge links it
        if not os.path.exists(_build.BRIDGE):
            raise RuntimeError(
                "pytorch_points_amd: %s not found. Build it with `python -m pytorch_points_amd._build` (g++ against "
                "the installed torch)." % _build.BRIDGE)
        import importlib
        if _build.bridge_abi_mismatch():   # (the tag only: file times do not survive a copy of the tree)
            raise RuntimeError(
                "pytorch_points_amd: %s was built for another torch / Python than this one (%s). Rebuild it with "
                "`python -m pytorch_points_amd._build`." % (_build.BRIDGE, _build.BRIDGE_META))
        try:
            mod = importlib.import_module("pytorch_points_amd._pp_torch")
        except ImportError as exc:
            raise RuntimeError("pytorch_points_amd: %s does not load (%s). Rebuild it with `python -m "
                               "pytorch_points_amd._build`." % (_build.BRIDGE, exc)) from exc
        mod.set_force_bruteforce(os.environ.get("PP_NMDISTANCE_SEARCH") == "bruteforce")
        _bridge = mod
    return _bridge


def check(code, what):
    if code != 0:
        raise RuntimeError("pytorch_points_amd: %s failed with HIP error %d" % (what, code))


PP_ENOTSUP = 801   # include/pp_hip.h: an *_ordered_* entry point cannot serve this shape


def deterministic():
    """torch.use_deterministic_algorithms(True): the scatter-add backward passes take their ordered forms
    (ascending source order, no floating-point atomics: reproducible bit for bit, equal to the CPU oracle)"""
    return torch.are_deterministic_algorithms_enabled()


def ordered_or_fallback(code, what):
    """outcome of an *_ordered_* call: True = done; False = this shape has no ordered form and torch is in
    warn-only mode (the caller takes the default path); raises otherwise"""
    if code == 0:
        return True
    if code != PP_ENOTSUP:
        check(code, what)
    msg = ("pytorch_points_amd: %s has no deterministic implementation for this shape "
           "(torch.use_deterministic_algorithms(True) is set)" % what)
    if torch.is_deterministic_algorithms_warn_only_enabled():
        import warnings
        warnings.warn(msg)
        return False
    raise RuntimeError(msg)


def require_cuda(*named):
    """The reference's CHECK_CUDA (_ext/utils.h:5): every tensor on a GPU, all on the same one."""
    dev = None
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("%s is on %s, expected %s" % (name, t.device, dev))
    return dev


def require_contiguous(*named):
    """CHECK_CONTIGUOUS (_ext/utils.h:6)"""
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % name)


def require_float(*named):
    """CHECK_IS_FLOAT (_ext/utils.h:15-19); fp32 is the only floating type of the sampling operators."""
    for name, t in named:
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be a float tensor" % name)


def require_float_or_double(*named):
    """The Chamfer operators' dtype rule: the reference dispatches them over float, double and half
    (AT_DISPATCH_FLOATING_TYPES_AND_HALF, _ext/nmdistance_cuda.cu:125); fp32 is the tuned path, fp64 and fp16 go through
    every-pair scans with the reference's arithmetic for the type; all floating arguments of one call in the same type.
    bfloat16 is not in that dispatch and raises TypeError.  -> the common dtype"""
    dt = named[0][1].dtype
    if dt is torch.bfloat16:
        raise TypeError("%s is %s: the Chamfer operators serve float32, float64 and float16 (the reference's dispatch)" % (named[0][0], dt))
    if dt not in (torch.float32, torch.float64, torch.float16):
        raise RuntimeError("%s must be a float tensor" % named[0][0])
    for name, t in named[1:]:
        if t.dtype != dt:
            if t.dtype is torch.bfloat16:
                raise TypeError("%s is %s: the Chamfer operators serve float32, float64 and float16" % (name, t.dtype))
            raise RuntimeError("%s must be a %s tensor like %s" % (name, "double" if dt is torch.float64 else "float", named[0][0]))
    return dt


def require_int(*named):
    """CHECK_IS_INT (_ext/utils.h:9-13)"""
    for name, t in named:
        if t.dtype != torch.int32:
            raise RuntimeError("%s must be an int tensor" % name)


def ptr(t):
    return _c_void_p(t.data_ptr())


# the cheap forms of torch.cuda.current_device() / is_current_stream_capturing() / current_stream().cuda_stream
# (the public wrappers cost a microsecond each; an operator call makes several)
try:
    current_device = torch._C._cuda_getDevice
    is_capturing = torch._C._cuda_isCurrentStreamCapturing
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # private API moved: the public one is slower, not different
    current_device = torch.cuda.current_device
    is_capturing = torch.cuda.is_current_stream_capturing
    _raw_stream = None


def raw_stream(device):
    """handle (int) of the current HIP stream of ``device`` -- the cheap form of
    torch.cuda.current_stream(device).cuda_stream"""
    idx = device.index if device.index is not None else current_device()
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


class on_device(object):
    """Device guard + current stream handle for a launch.  The guard is skipped when the tensors'
    device is already current (the common case; entering a guard costs several microseconds)."""

    def __init__(self, device):
        self.device = device
        self._guard = None

    def __enter__(self):
        idx = self.device.index
        if idx is not None and idx != current_device():
            self._guard = torch.cuda.device(self.device)
            self._guard.__enter__()
        return _c_void_p(raw_stream(self.device))

    def __exit__(self, *exc):
        if self._guard is not None:
            return self._guard.__exit__(*exc)
        return False


# ------------------------------------------------------------------------------------------ scratch
# One growing uint8 buffer per (device, stream, operator): calls on one stream are ordered, calls on
# different streams must not share scratch.  Least recently used entries are dropped beyond _WS_MAX
# (a buffer is ~40 MB at Chamfer config 2).
_WS = collections.OrderedDict()
_WS_MAX = 16


def workspace(device, name, nbytes):
    """Scratch of at least ``nbytes`` for operator ``name`` on the current stream of ``device`` (None
    for 0 bytes).  While the stream is being captured into a graph (``torch.cuda.graph``) nothing is
    cached: the buffer comes from the capturing graph's private pool, serves the launches being
    recorded and is released by the caller, so every graph owns its scratch and no block of a graph's
    pool outlives the capture in this table (two graphs replayed on different streams, or a replay
    beside an eager call, would otherwise share one buffer)."""
    if not nbytes:
        return None
    if is_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (device.index, raw_stream(device), name)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WS[key] = buf
        while len(_WS) > _WS_MAX:
            _WS.popitem(last=False)
    else:
        _WS.move_to_end(key)
    return buf


def cached_workspaces(name, device=None):
    """the cached scratch buffers of operator ``name`` (tests and debug aids)"""
    idx = None if device is None else (device.index if device.index is not None else current_device())
    return [buf for (d, _, n), buf in _WS.items() if n == name and (idx is None or d is None or d == idx)]
