"""``pytorch_points._ext.sampling`` (reference: _ext/sampling.cpp:205-216).

Same nine names and positional signatures as the pybind module.  Ownership follows the reference:
``ball_query``, ``group_points`` and ``group_points_grad`` allocate and return their output on the
input's device (sampling.cpp:93-94,123-125,148-150); everything else writes into caller-allocated
tensors.  Precondition failures raise RuntimeError (the reference's AT_ASSERTM / TORCH_CHECK);
launch failures raise RuntimeError instead of the reference's ``exit(-1)``.
"""
import torch

from .. import _lib

# (device index, raw stream) -> [pinned int32[1], event]: host mirror of the sticky status word of THAT stream's FPS
# workspace (one workspace per device and stream: _lib.workspace), and an event recorded behind the copy into it
_fps_status_mirror = {}
_FPS_TIMEOUT = ("pytorch_points_amd: an earlier furthest_sampling call on %s timed out waiting for its workgroups to be "
                "co-resident (another stream kept the GPU busy); its indices are zeros from the failing step on.  "
                "Re-run it, or serialise it with the other stream.")


def _fps_clear(dev, key, entry):
    """forget a reported timeout: the mirror, and the status word of the workspace it mirrors (on that workspace's
    own stream: the current one -- the key is (device, current stream))"""
    entry[0].zero_()
    ws = _lib._WS.get((key[0], key[1], "fps"))   # (keyed as _lib.workspace keys it: the device's integer index --
    #                                                  `dev` may carry none)
    if ws is not None:
        ws[:4].zero_()


def furthest_sampling_check(device=None):
    """Explicit check for callers that want to know at the end of a step: waits for the last
    furthest_sampling call issued on the CURRENT stream of ``device`` (an event behind its status copy -- not a
    device synchronisation) and raises RuntimeError if one of that stream's calls timed out since the last report.
    Without it a timeout is reported by the next furthest_sampling call on the same device and stream; a call that
    was captured into a graph (torch.cuda.graph) never reports one: there is no host code in a replay."""
    dev = torch.device("cuda", _lib.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else _lib.current_device()
    key = (idx, _lib.raw_stream(dev))
    entry = _fps_status_mirror.get(key)
    if entry is None:
        return
    entry[1].synchronize()
    if int(entry[0][0]) != 0:
        with torch.cuda.device(idx):
            _fps_clear(dev, key, entry)
        raise RuntimeError(_FPS_TIMEOUT % dev)


def _workspace(device, nbytes):
    return _lib.workspace(device, "fps", nbytes)


def _scatter_ws(device, b, triples, destinations, per_source, weighted):
    """(buffer, bytes) of the per-(device, stream) scratch for the sorted scatter-add; (None, 0) when
    the problem does not qualify.  The caller keeps the buffer alive until its launch is issued."""
    nbytes = int(_lib.lib().pp_scatter_workspace_bytes(b, triples, destinations, per_source, weighted))
    if nbytes == 0:
        return None, 0
    return _lib.workspace(device, "scatter", nbytes), nbytes


def furthest_sampling_status(device):
    """Test/debug aid: 0 if no inter-workgroup wait of the FPS cluster kernel has timed out on this
    device's workspace since its last call (synchronises the current stream)."""
    device = torch.device(device)
    status = 0
    for ws in _lib.cached_workspaces("fps", device):
        with _lib.on_device(device) as stream:
            status |= int(_lib.lib().pp_furthest_sampling_status(_lib.ptr(ws), stream))
    return status


def furthest_sampling(m, seedIdx, input, temp, idx, sampled=None, channels_first=False):
    """furthest_sampling_forward (sampling.cpp:68-82): input (B,N,3), temp (B,N) in/out, idx (B,m)
    out; returns idx.  ``temp=None``: every point starts at 1e10 and nothing is kept (what the reference's wrapper does
    with a temp of its own).  Beyond the reference's signature: ``sampled`` (B,m,3) -- or (B,3,m) with
    ``channels_first`` -- receives the picked points' coordinates in the same launch (what
    network/geo_operations.py:59-63 does with a gather_points call afterwards)."""
    if temp is None:   # (beyond the reference's signature: "a temp of the callee's own" -- see below)
        dev = _lib.require_cuda(("input", input), ("idx", idx))
        _lib.require_contiguous(("input", input), ("idx", idx))
        _lib.require_float(("input", input))
    else:
        dev = _lib.require_cuda(("input", input), ("temp", temp), ("idx", idx))
        _lib.require_contiguous(("input", input), ("temp", temp), ("idx", idx))  # CHECK_INPUT :76-77
        _lib.require_float(("input", input), ("temp", temp))
    _lib.require_int(("idx", idx))
    if input.dim() != 3 or input.size(2) != 3:
        raise RuntimeError("input must be (B, N, 3)")
    b, n, _ = input.shape
    if (temp is not None and temp.numel() != b * n) or idx.numel() != b * int(m):
        raise RuntimeError("temp must be (B, N) and idx (B, m)")
    if sampled is not None:
        _lib.require_cuda(("input", input), ("sampled", sampled))
        _lib.require_contiguous(("sampled", sampled))
        _lib.require_float(("sampled", sampled))
        if tuple(sampled.shape) != ((b, 3, int(m)) if channels_first else (b, int(m), 3)):
            raise RuntimeError("sampled must be (B, m, 3), or (B, 3, m) with channels_first")
    L = _lib.lib()
    nbytes = int(L.pp_furthest_sampling_workspace_bytes(b, n, int(m)))
    ws = _workspace(dev, nbytes)
    capturing = _lib.is_capturing()
    didx = dev.index if dev.index is not None else _lib.current_device()
    key = (didx, _lib.raw_stream(dev))
    entry = _fps_status_mirror.get(key)
    if entry is not None and not capturing and int(entry[0][0]) != 0:
        # The status word of an EARLIER call on this device and stream, copied to pinned host memory behind that call
        # (no synchronisation here): a bounded wait between the workgroups of the cluster kernel timed out, e.g.
        # because a kernel of another stream held the CUs its members were waiting for.  That call left zeros in its
        # indices.  (furthest_sampling_check() reports it without waiting for the next call.)
        with torch.cuda.device(didx):
            _fps_clear(dev, key, entry)
        raise RuntimeError(_FPS_TIMEOUT % dev)
    if ws is not None and not getattr(ws, "_pp_status_zeroed", False):
        ws[:256].zero_()             # the sticky status word in front of the scratch (include/pp_hip.h)
        ws._pp_status_zeroed = True
    with _lib.on_device(dev) as stream:
        def call(t):
            return L.pp_furthest_sampling_gather_f32(
                _lib.ptr(input), _lib.ptr(t) if t is not None else None, _lib.ptr(idx),
                _lib.ptr(sampled) if sampled is not None else None, 1 if channels_first else 0, b, n, int(m),
                int(seedIdx), _lib.ptr(ws) if ws is not None else None, nbytes, stream)
        code = call(temp)
        if temp is None and code == _lib.PP_ENOTSUP:
            # temp=None ("start at 1e10, keep nothing": network/geo_operations.py:33 fills a temp nobody reads back) is
            # served where the bucketed kernel runs; the other kernels need a buffer
            code = call(torch.full((b, n), 1e10, dtype=torch.float32, device=dev))
        _lib.check(code, "furthest_sampling")
        if ws is not None and not capturing:
            if entry is None:
                entry = _fps_status_mirror[key] = [torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event()]
                while len(_fps_status_mirror) > 64:      # (streams come and go: oldest first)
                    _fps_status_mirror.pop(next(iter(_fps_status_mirror)))
            # the word is sticky on the device (set by a timed-out wait, cleared only by _fps_clear), so the copy of a
            # later clean call on the same workspace still carries an unreported 1; other streams have other
            # workspaces and other mirrors.  Looked at by the NEXT call or by furthest_sampling_check, never waited
            # for here.
            entry[0].copy_(ws[:4].view(torch.int32), non_blocking=True)
            entry[1].record()
    return idx


def gather_forward(b, c, n, npoints, points, idx, out):
    """gather_points_wrapper_fast (sampling.cpp:19-28): out[b,c,m] = points[b,c,idx[b,m]]"""
    dev = _lib.require_cuda(("points", points), ("idx", idx), ("out", out))
    _lib.require_contiguous(("points", points), ("idx", idx), ("out", out))
    _lib.require_float(("points", points), ("out", out))
    _lib.require_int(("idx", idx))
    if points.numel() != b * c * n or idx.numel() != b * npoints or out.numel() != b * c * npoints:
        raise RuntimeError("gather_forward: tensor sizes do not match (b, c, n, npoints)")
    with _lib.on_device(dev) as stream:
        _lib.check(_lib.lib().pp_gather_forward_f32(
            _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out), b, c, n, npoints, stream),
            "gather_forward")
    return 1


def gather_backward(b, c, n, npoints, grad_out, idx, grad_points):
    """gather_points_grad_wrapper_fast (sampling.cpp:31-41): scatter-add into grad_points."""
    dev = _lib.require_cuda(("grad_out", grad_out), ("idx", idx), ("grad_points", grad_points))
    _lib.require_contiguous(("grad_out", grad_out), ("idx", idx), ("grad_points", grad_points))
    _lib.require_float(("grad_out", grad_out), ("grad_points", grad_points))
    _lib.require_int(("idx", idx))
    if (grad_out.numel() != b * c * npoints or idx.numel() != b * npoints
            or grad_points.numel() != b * c * n):
        raise RuntimeError("gather_backward: tensor sizes do not match (b, c, n, npoints)")
    with _lib.on_device(dev) as stream:
        ws, nbytes = _scatter_ws(dev, b, npoints, n, 1, 0)
        args = (_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(grad_points), b, c, n, npoints,
                _lib.ptr(ws) if ws is not None else None, nbytes, stream)
        L = _lib.lib()
        if not (_lib.deterministic() and _lib.ordered_or_fallback(L.pp_gather_backward_ordered_f32(*args), "gather_backward")):
            _lib.check(L.pp_gather_backward_ws_f32(*args), "gather_backward")
    return 1


def _named_workspace(dev, name, nbytes):
    """scratch for the grid searches, one per (device, stream, op); None when nbytes == 0"""
    return _lib.workspace(dev, name, nbytes)


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query_wrapper_fast (sampling.cpp:85-104): new_xyz (B,M,3) centres, xyz (B,N,3) ->
    int32 idx (B,M,nsample)."""
    dev = _lib.require_cuda(("new_xyz", new_xyz), ("xyz", xyz))
    _lib.require_contiguous(("new_xyz", new_xyz), ("xyz", xyz))
    _lib.require_float(("new_xyz", new_xyz), ("xyz", xyz))
    if new_xyz.dim() != 3 or xyz.dim() != 3 or new_xyz.size(2) != 3 or xyz.size(2) != 3 \
            or new_xyz.size(0) != xyz.size(0):
        raise RuntimeError("new_xyz must be (B, M, 3) and xyz (B, N, 3)")
    b, m, _ = new_xyz.shape
    n = xyz.size(1)
    nsample = int(nsample)
    idx = torch.empty(b, m, nsample, dtype=torch.int32, device=dev)  # kernel writes every slot
    with _lib.on_device(dev) as stream:
        nbytes = int(_lib.lib().pp_ball_query_workspace_bytes(b, n, m, nsample))
        ws = _named_workspace(dev, "ball_query", nbytes)
        _lib.check(_lib.lib().pp_ball_query_ws_f32(
            _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx), b, n, m, float(radius), nsample,
            _lib.ptr(ws) if ws is not None else None, nbytes, stream), "ball_query")
    return idx


def group_points(points, idx):
    """group_points (sampling.cpp:113-138): points (B,C,N), idx (B,npoint,nsample) ->
    (B,C,npoint,nsample)."""
    _lib.require_contiguous(("points", points), ("idx", idx))
    _lib.require_float(("points", points))
    _lib.require_int(("idx", idx))
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")  # sampling.cpp:132
    dev = _lib.require_cuda(("points", points), ("idx", idx))
    if points.dim() != 3 or idx.dim() != 3 or points.size(0) != idx.size(0):
        raise RuntimeError("points must be (B, C, N) and idx (B, npoint, nsample)")
    b, c, n = points.shape
    _, npoint, nsample = idx.shape
    out = torch.empty(b, c, npoint, nsample, dtype=torch.float32, device=dev)  # fully written
    with _lib.on_device(dev) as stream:
        _lib.check(_lib.lib().pp_group_points_f32(
            _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out), b, c, n, npoint, nsample, stream),
            "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """group_points_grad (sampling.cpp:140-161): grad_out (B,C,npoint,nsample) -> (B,C,n)."""
    _lib.require_contiguous(("grad_out", grad_out), ("idx", idx))
    _lib.require_float(("grad_out", grad_out))
    _lib.require_int(("idx", idx))
    if not grad_out.is_cuda:
        raise RuntimeError("CPU not supported")  # sampling.cpp:157
    dev = _lib.require_cuda(("grad_out", grad_out), ("idx", idx))
    if grad_out.dim() != 4 or idx.dim() != 3:
        raise RuntimeError("grad_out must be (B, C, npoint, nsample) and idx (B, npoint, nsample)")
    b, c, npoint, nsample = grad_out.shape
    # (the deterministic form accumulates into zeros, as the reference's does; the default form WRITES every element:
    #  no fill in front of it, no read of the output -- pp_group_points_grad_out_ws_f32)
    det = _lib.deterministic()
    out = (torch.zeros if det else torch.empty)(b, c, int(n), dtype=torch.float32, device=dev)
    with _lib.on_device(dev) as stream:
        ws, nbytes = _scatter_ws(dev, b, npoint * nsample, int(n), 1, 0)
        args = (_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(out), b, c, int(n), npoint, nsample,
                c * npoint * nsample, _lib.ptr(ws) if ws is not None else None, nbytes, stream)
        L = _lib.lib()
        if not (det and _lib.ordered_or_fallback(L.pp_group_points_grad_ordered_f32(*args), "group_points_grad")):
            _lib.check(L.pp_group_points_grad_out_ws_f32(*args), "group_points_grad")
    return out


# --- extensions (not in the reference's module): the gather / its gradient on a channel slice of a
# --- wider tensor, so that QueryAndGroup can write its concatenated output once -------------------
def group_points_into(points, idx, out, channel_offset):
    """out[:, channel_offset : channel_offset + C] = group_points(points, idx), written in place.
    points (B,C,N), idx (B,npoint,nsample), out (B,Ctot,npoint,nsample) contiguous."""
    _lib.require_contiguous(("points", points), ("idx", idx), ("out", out))
    _lib.require_float(("points", points), ("out", out))
    _lib.require_int(("idx", idx))
    dev = _lib.require_cuda(("points", points), ("idx", idx), ("out", out))
    b, c, n = points.shape
    _, npoint, nsample = idx.shape
    if out.dim() != 4 or out.size(0) != b or out.size(2) != npoint or out.size(3) != nsample \
            or channel_offset < 0 or channel_offset + c > out.size(1):
        raise RuntimeError("out must be (B, Ctot, npoint, nsample) with room for C channels at channel_offset")
    p = npoint * nsample
    with _lib.on_device(dev) as stream:
        _lib.check(_lib.lib().pp_group_points_strided_f32(
            _lib.ptr(points), _lib.ptr(idx), _lib._c_void_p(out.data_ptr() + 4 * channel_offset * p),
            b, c, n, npoint, nsample, out.size(1) * p, stream), "group_points_into")
    return out


def group_points_grad_from(grad_out, idx, n, channel_offset, channels):
    """group_points_grad of grad_out[:, channel_offset : channel_offset + channels] -> (B,channels,n),
    without materialising the slice.  grad_out (B,Ctot,npoint,nsample) contiguous."""
    _lib.require_contiguous(("grad_out", grad_out), ("idx", idx))
    _lib.require_float(("grad_out", grad_out))
    _lib.require_int(("idx", idx))
    dev = _lib.require_cuda(("grad_out", grad_out), ("idx", idx))
    b, ctot, npoint, nsample = grad_out.shape
    if channel_offset < 0 or channel_offset + channels > ctot:
        raise RuntimeError("channel slice out of range")
    p = npoint * nsample
    det = _lib.deterministic()
    out = (torch.zeros if det else torch.empty)(b, channels, int(n), dtype=torch.float32, device=dev)
    with _lib.on_device(dev) as stream:
        ws, nbytes = _scatter_ws(dev, b, p, int(n), 1, 0)
        args = (_lib._c_void_p(grad_out.data_ptr() + 4 * channel_offset * p), _lib.ptr(idx), _lib.ptr(out),
                b, channels, int(n), npoint, nsample, ctot * p, _lib.ptr(ws) if ws is not None else None, nbytes, stream)
        L = _lib.lib()
        if not (det and _lib.ordered_or_fallback(L.pp_group_points_grad_ordered_f32(*args), "group_points_grad")):
            _lib.check(L.pp_group_points_grad_out_ws_f32(*args), "group_points_grad_from")
    return out


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """three_nn_wrapper_fast (sampling.cpp:163-172): fills dist2 (B,N,3) squared, idx (B,N,3)."""
    floats = (("unknown", unknown), ("known", known), ("dist2", dist2))
    dev = _lib.require_cuda(*floats, ("idx", idx))
    _lib.require_contiguous(*floats, ("idx", idx))
    _lib.require_float(*floats)
    _lib.require_int(("idx", idx))
    if unknown.numel() != b * n * 3 or known.numel() != b * m * 3 or dist2.numel() != b * n * 3 \
            or idx.numel() != b * n * 3:
        raise RuntimeError("three_nn_wrapper: tensor sizes do not match (b, n, m)")
    with _lib.on_device(dev) as stream:
        nbytes = int(_lib.lib().pp_three_nn_workspace_bytes(b, n, m))
        ws = _named_workspace(dev, "three_nn", nbytes)
        _lib.check(_lib.lib().pp_three_nn_ws_f32(
            _lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2), _lib.ptr(idx), b, n, m,
            _lib.ptr(ws) if ws is not None else None, nbytes, stream), "three_nn_wrapper")


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    """three_interpolate_wrapper_fast (sampling.cpp:175-188): points (B,C,M), idx/weight (B,N,3)
    -> out (B,C,N)."""
    floats = (("points", points), ("weight", weight), ("out", out))
    dev = _lib.require_cuda(*floats, ("idx", idx))
    _lib.require_contiguous(*floats, ("idx", idx))
    _lib.require_float(*floats)
    _lib.require_int(("idx", idx))
    if points.numel() != b * c * m or idx.numel() != b * n * 3 or weight.numel() != b * n * 3 \
            or out.numel() != b * c * n:
        raise RuntimeError("three_interpolate_wrapper: tensor sizes do not match (b, c, m, n)")
    with _lib.on_device(dev) as stream:
        _lib.check(_lib.lib().pp_three_interpolate_f32(
            _lib.ptr(points), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(out), b, c, m, n, stream),
            "three_interpolate_wrapper")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    """three_interpolate_grad_wrapper_fast (sampling.cpp:190-203): scatter-add into grad_points
    (B,C,M)."""
    floats = (("grad_out", grad_out), ("weight", weight), ("grad_points", grad_points))
    dev = _lib.require_cuda(*floats, ("idx", idx))
    _lib.require_contiguous(*floats, ("idx", idx))
    _lib.require_float(*floats)
    _lib.require_int(("idx", idx))
    if grad_out.numel() != b * c * n or idx.numel() != b * n * 3 or weight.numel() != b * n * 3 \
            or grad_points.numel() != b * c * m:
        raise RuntimeError("three_interpolate_grad_wrapper: tensor sizes do not match (b, c, n, m)")
    with _lib.on_device(dev) as stream:
        ws, nbytes = _scatter_ws(dev, b, 3 * n, m, 3, 1)
        args = (_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(grad_points), b, c, n, m,
                _lib.ptr(ws) if ws is not None else None, nbytes, stream)
        L = _lib.lib()
        if not (_lib.deterministic() and _lib.ordered_or_fallback(L.pp_three_interpolate_grad_ordered_f32(*args),
                                                                  "three_interpolate_grad")):
            _lib.check(L.pp_three_interpolate_grad_ws_f32(*args), "three_interpolate_grad_wrapper")
