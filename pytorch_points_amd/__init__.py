"""pytorch_points_amd -- MI355X (gfx950) implementation of the pytorch_points `_ext` hot path.

Drop-in for the reference's operator API on that path only:
    pytorch_points.network.model_loss.{nndistance, labeled_nndistance}
    pytorch_points.network.operations.{gather_points, ball_query, grouping_operation, QueryAndGroup}
    pytorch_points.network.geo_operations.furthest_point_sample
    pytorch_points.network.pointnet2_utils.{three_nn, three_interpolate, QueryAndGroup, GroupAll}
    pytorch_points._ext.{losses, sampling}
Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the kernels are
hand-written HIP in csrc/, reached through the C ABI of libpp_hip.so (include/pp_hip.h).
"""
import torch as _torch

# the reference sets this on import (pytorch_points/__init__.py:1-2)
_torch.backends.cudnn.benchmark = False

__version__ = "0.1.0"


def install_as_pytorch_points():
    """Register this package under the reference's import name, so that
    ``from pytorch_points.network.model_loss import nndistance`` resolves here."""
    import importlib
    import sys
    names = ["", "._ext", "._ext.losses", "._ext.sampling", ".network", ".network.model_loss",
             ".network.operations", ".network.geo_operations", ".network.pointnet2_utils"]
    for suffix in names:
        mod = importlib.import_module(__name__ + suffix)
        sys.modules["pytorch_points" + suffix] = mod


def install_knn_as_pytorch3d_ops():
    """Opt-in: make ``import pytorch3d.ops as ops; ops.knn_points(...)`` -- the only use the reference
    makes of pytorch3d -- resolve to pytorch_points_amd.ops when pytorch3d itself is not installed.
    Does nothing if a real pytorch3d is importable."""
    import importlib
    import sys
    import types
    try:
        importlib.import_module("pytorch3d.ops")
        return False
    except ImportError:
        pass
    from . import ops
    pkg = types.ModuleType("pytorch3d")
    pkg.__path__ = []
    pkg.ops = ops
    sys.modules["pytorch3d"] = pkg
    sys.modules["pytorch3d.ops"] = ops
    return True
