"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/pp_hip.h declares;
the host wrappers reject what the reference's precondition macros reject; the drop-in import
names resolve; there is no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

from pytorch_points_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="pp_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_whole_path():
    syms = declared_symbols()
    for s in ["pp_nmdistance_forward_f32", "pp_labeled_nmdistance_forward_f32", "pp_nmdistance_backward_f32",
              "pp_furthest_sampling_f32", "pp_gather_forward_f32", "pp_gather_backward_f32", "pp_ball_query_f32",
              "pp_group_points_f32", "pp_group_points_grad_f32", "pp_three_nn_f32", "pp_three_interpolate_f32",
              "pp_three_interpolate_grad_f32"]:
        assert s in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    _build.build()
    assert os.path.exists(_build.LIB)
    handle = ctypes.CDLL(_build.LIB)
    for s in declared_symbols():
        assert hasattr(handle, s), "libpp_hip.so does not export %s" % s
    # and the ctypes table binds exactly the declared functions
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.version().startswith("pp_hip") and "gfx950" in _lib.version()


def test_debug_knobs_are_declared_and_nothing_else_is_exported():
    """every exported pp_* symbol is declared in include/pp_hip.h (the drop-in ABI) or
    include/pp_hip_debug.h (test / benchmark knobs); the library exports no undeclared entry point."""
    import subprocess
    _build.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _build.LIB], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("pp_")})
    declared = sorted(set(declared_symbols()) | set(declared_symbols("pp_hip_debug.h")))
    assert exported == declared, (sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported)))
    assert all(s.startswith("pp_debug_") for s in declared_symbols("pp_hip_debug.h"))
    assert not any(s.startswith("pp_debug_") for s in declared_symbols())


def test_product_code_never_touches_a_debug_knob():
    pkg = os.path.join(ROOT, "pytorch_points_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "graphs.py":
                assert "pp_debug_" not in open(os.path.join(d, f)).read(), os.path.join(d, f)


def test_reference_import_name_resolves_without_an_install_call():
    """``import pytorch_points`` (repo root on sys.path) is this package: the reference's hot-path import lines
    work as written."""
    import subprocess
    import sys
    code = ("import pytorch_points, pytorch_points_amd;"
            "from pytorch_points.network.model_loss import nndistance, labeled_nndistance;"
            "from pytorch_points.network.operations import QueryAndGroup, gather_points, ball_query, grouping_operation;"
            "from pytorch_points.network.geo_operations import furthest_point_sample;"
            "from pytorch_points.network.pointnet2_utils import three_nn, three_interpolate, GroupAll;"
            "from pytorch_points._ext import losses, sampling;"
            "import pytorch_points_amd.network.model_loss as m;"
            "assert nndistance is m.nndistance and pytorch_points is pytorch_points_amd;print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_native_autograd_bridge_builds_and_loads():
    """csrc/torch_bridge.cpp -> _pp_torch.so (g++ against the installed torch); it links the C-ABI library."""
    _build.build()
    assert os.path.exists(_build.BRIDGE)
    b = _lib.bridge()
    assert b.library_version() == _lib.version()
    assert callable(b.nndistance) and callable(b.labeled_nndistance)


def test_code_object_is_gfx950_only():
    blob = open(_build.LIB, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80"):
        assert other not in blob


def test_opt_n_threads_matches_reference_helper():
    # _ext/cuda_utils.h:11-16
    import oracle
    L = _lib.lib()
    for n in list(range(1, 70)) + [127, 128, 255, 256, 300, 511, 512, 513, 1023, 1024, 5000, 65536, 1 << 20, (1 << 29), (1 << 29) + 1]:
        assert L.pp_opt_n_threads(n) == oracle.opt_n_threads(n)
    assert L.pp_opt_n_threads(2048) == 512 and L.pp_opt_n_threads(300) == 256 and L.pp_opt_n_threads(1) == 1


def test_workspace_size_queries_are_host_only():
    """The two *_workspace_bytes entry points are pure host arithmetic (callable without a GPU)."""
    L = _lib.lib()
    # grid search: only C == 3 and clouds of >= 2048 points; 0 means "brute force only"
    assert L.pp_nmdistance_forward_workspace_bytes(32, 16384, 16384, 3) > 20e6
    assert L.pp_nmdistance_forward_workspace_bytes(2, 1024, 1024, 3) == 0
    assert L.pp_nmdistance_forward_workspace_bytes(2, 4096, 4096, 5) == 0
    assert L.pp_nmdistance_forward_workspace_bytes(0, 4096, 4096, 3) == 0
    # FPS cluster ring: B * CL <= 256 co-resident workgroups, >= 512 points each
    assert L.pp_furthest_sampling_workspace_bytes(16, 65536, 4096) > 0
    assert L.pp_furthest_sampling_workspace_bytes(300, 1024, 64) == 0      # more batch elements than CUs, small clouds
    # the bucketed kernel (N >= 2048, 32 or more picks): the sorted cloud (16 B per point) and one more word per point
    # (up to 65536 points: a word for every register slot of the running minima, 65536 per batch element)
    assert L.pp_furthest_sampling_workspace_bytes(300, 2048, 64) == 256 + 300 * (2048 * 16 + 65536 * 4)
    # from 32768 points the counting sort runs as launches of their own: its tables (box parts, three 1024-bin axis
    # histograms, 32768 cell counters) per batch element
    assert L.pp_furthest_sampling_workspace_bytes(16, 65536, 4096) > 16 * (65536 * 20 + 32768 * 4 + 3 * 1024 * 4)
    assert L.pp_furthest_sampling_workspace_bytes(1, 70000, 64) >= 256 + 70000 * 20
    assert L.pp_furthest_sampling_workspace_bytes(1, 600, 64) == 0         # too small to split
    # argument validation happens before anything touches a device
    assert L.pp_nmdistance_forward_f32(None, None, None, None, None, None, -1, 4, 4, 3, None) != 0
    assert L.pp_nmdistance_forward_f32(None, None, None, None, None, None, 0, 4, 4, 3, None) == 0
    assert L.pp_ball_query_f32(None, None, None, 2, 8, 0, 0.1, 4, None) == 0
    assert L.pp_furthest_sampling_f32(None, None, None, 2, 8, 0, 0, None, 0, None) == 0   # npoint <= 0: no-op
    assert L.pp_furthest_sampling_f32(None, None, None, 2, 8, 4, 99, None, 0, None) != 0  # null / bad seed


def test_no_cpu_fallback():
    from pytorch_points_amd.network.model_loss import nndistance, labeled_nndistance
    from pytorch_points_amd.network.operations import gather_points, ball_query, grouping_operation
    from pytorch_points_amd.network.geo_operations import furthest_point_sample
    from pytorch_points_amd.network.pointnet2_utils import three_nn, three_interpolate
    x = torch.zeros(1, 8, 3)
    i = torch.zeros(1, 4, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        nndistance(x, x)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        labeled_nndistance(x, x, torch.zeros(1, 8), torch.zeros(1, 8))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        gather_points(x.transpose(1, 2).contiguous(), i)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ball_query(0.1, 4, x, x)
    with pytest.raises(RuntimeError, match="CPU not supported"):     # sampling.cpp:131-133
        grouping_operation(x.transpose(1, 2).contiguous(), torch.zeros(1, 2, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        furthest_point_sample(x, 4, NCHW=False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        three_nn(x, x)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        three_interpolate(x.transpose(1, 2).contiguous(), torch.zeros(1, 4, 3, dtype=torch.int32), torch.zeros(1, 4, 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pytorch_points_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "pp_oracle" not in text, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_build, "LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_drop_in_import_names():
    import pytorch_points_amd
    pytorch_points_amd.install_as_pytorch_points()
    from pytorch_points.network.model_loss import nndistance, labeled_nndistance, NmDistanceFunction  # noqa: F401
    from pytorch_points.network.operations import gather_points, ball_query, grouping_operation, QueryAndGroup  # noqa: F401
    from pytorch_points.network.geo_operations import furthest_point_sample  # noqa: F401
    from pytorch_points.network.pointnet2_utils import three_nn, three_interpolate, GroupAll, QueryAndGroup as Q2  # noqa: F401
    from pytorch_points._ext import losses, sampling
    for name in ["nmdistance_forward", "labeled_nmdistance_forward", "nmdistance_backward"]:   # nmdistance.cpp:30-34
        assert callable(getattr(losses, name))
    for name in ["furthest_sampling", "gather_forward", "gather_backward", "ball_query", "group_points",
                 "group_points_grad", "three_nn_wrapper", "three_interpolate_wrapper",
                 "three_interpolate_grad_wrapper"]:                                            # sampling.cpp:205-216
        assert callable(getattr(sampling, name))
    assert torch.backends.cudnn.benchmark is False      # pytorch_points/__init__.py:2
